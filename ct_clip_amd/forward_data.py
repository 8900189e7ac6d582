"""Latent export (drop-in for ``scripts/forward_data.py``: ``CTClipInference``).

For every (volume, report) of the dataset the reference calls ``model(text_tokens, volume, return_latents=True)`` and stores the
text latent under ``text/<accession>.npz`` and the image latent under ``image/<accession>.npz`` (key ``arr``; forward_data.py:131-148)
for the retrieval scripts.  Same files here, produced by the HIP towers (``CTCLIP.forward(return_latents=True)``)."""
from pathlib import Path

import torch
from torch import nn

from . import distributed as _dist


class CTClipInference(nn.Module):
    """Additive kwargs: ``dataset`` (indexable, yields ``(volume, report_text, onehot_labels, accession_name)``; the reference's NIfTI
    dataset is only built when it is None and ``data_inference`` is importable) and ``tokenizer``."""

    def __init__(self, CTClip, *, data_folder="external_valid", reports_file="data_reports.xslx", meta_file="meta_data.csv",
                 results_folder="./results", labels="labels.csv", accelerate_kwargs: dict = dict(), dataset=None, tokenizer=None,
                 max_text_len=512):
        super().__init__()
        self.CTClip = CTClip
        if tokenizer is None:
            from transformers import BertTokenizer
            tokenizer = BertTokenizer.from_pretrained("microsoft/BiomedVLP-CXR-BERT-specialized", do_lower_case=True)
        self.tokenizer = tokenizer
        self.max_text_len = max_text_len
        self.register_buffer("steps", torch.Tensor([0]))
        if dataset is None:
            from data_inference import CTReportDatasetinfer   # the reference's scripts/data_inference.py (needs nibabel)
            dataset = CTReportDatasetinfer(data_folder=data_folder, reports_file=reports_file, meta_file=meta_file, labels=labels)
        self.ds = dataset
        self.device = self.CTClip.temperature.device
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(parents=True, exist_ok=True)

    @property
    def is_main(self):
        return _dist.rank() == 0

    def print(self, msg):
        if self.is_main:
            print(msg)

    @torch.no_grad()
    def infer(self, log_fn=lambda *a, **k: None):
        import numpy as np
        model = self.CTClip
        model.eval()
        (self.results_folder / "text").mkdir(parents=True, exist_ok=True)
        (self.results_folder / "image").mkdir(parents=True, exist_ok=True)
        for i in range(len(self.ds)):
            volume, text, _onehot, acc = self.ds[i]
            volume = torch.as_tensor(volume)
            if volume.dim() == 4:
                volume = volume[None]
            name = acc if isinstance(acc, str) else acc[0]
            texts = [text] if isinstance(text, str) else list(text)
            tokens = self.tokenizer(texts, return_tensors="pt", padding="max_length", truncation=True,
                                    max_length=self.max_text_len).to(self.device)
            text_lat, image_lat, _tokens = model(tokens, volume.to(self.device), device=self.device, return_latents=True)
            np.savez(str(self.results_folder / "text" / f"{name}.npz"), arr=text_lat.float().cpu().numpy())
            np.savez(str(self.results_folder / "image" / f"{name}.npz"), arr=image_lat.float().cpu().numpy())
        log_fn({})
        self.print("Inference complete")
