"""Zero-shot pathology classification (drop-in for ``scripts/zero_shot.py``: ``CTClipInference``).

The reference scores one volume against 18 pathologies by calling the whole model 18 times with a (2-prompt, 1-volume) pair
(scripts/zero_shot.py:133-143): 18 image-tower passes of the SAME volume and 36 text passes of prompts that never change.  Here the
image tower runs once per volume and the 36 prompt latents are computed once per model: the scores are the no-loss similarity of
``CTCLIP.forward`` (ct_clip.py:805-807), ``softmax over the (present, absent) pair of temp * <text_latent, image_latent>``,
so the numbers are the reference's (checked against the pair-by-pair path in tests/test_zero_shot_cpu.py).
"""
from pathlib import Path

import torch
from torch import nn

from . import distributed as _dist

PATHOLOGIES = ['Medical material', 'Arterial wall calcification', 'Cardiomegaly', 'Pericardial effusion',
               'Coronary artery wall calcification', 'Hiatal hernia', 'Lymphadenopathy', 'Emphysema', 'Atelectasis', 'Lung nodule',
               'Lung opacity', 'Pulmonary fibrotic sequela', 'Pleural effusion', 'Mosaic attenuation pattern',
               'Peribronchial thickening', 'Consolidation', 'Bronchiectasis', 'Interlobular septal thickening']


def prompts_for(pathologies, present="{} is present.", absent="{} is not present."):
    """(2 * P) prompts, pair p at positions 2p (present) and 2p + 1 (absent) -- scripts/zero_shot.py:134."""
    out = []
    for name in pathologies:
        out += [present.format(name), absent.format(name)]
    return out


class ZeroShotClassifier:
    """Caches the prompt latents of one model; ``predict(volume)`` costs one image-tower pass."""

    def __init__(self, clip, tokenizer, pathologies=PATHOLOGIES, max_length=512, present="{} is present.",
                 absent="{} is not present.", text_batch=12):
        self.clip, self.tokenizer, self.pathologies = clip, tokenizer, list(pathologies)
        self.max_length, self.templates, self.text_batch = max_length, (present, absent), text_batch
        self._text = None

    def invalidate(self):
        """Call after the model's weights change (the cached prompt latents are a function of the text tower)."""
        self._text = None

    @torch.no_grad()
    def text_latents(self):
        """(P, 2, dim_latent) l2-normalised latents of the (present, absent) prompts, computed once."""
        if self._text is None:
            clip = self.clip
            was_training = clip.training
            clip.eval()
            dev = clip.temperature.device
            prompts = prompts_for(self.pathologies, *self.templates)
            chunks = []
            for i in range(0, len(prompts), self.text_batch):       # bounded text batches: 36 x 512 tokens at once is needless
                tok = self.tokenizer(prompts[i:i + self.text_batch], return_tensors="pt", padding="max_length", truncation=True,
                                     max_length=self.max_length).to(dev)
                chunks.append(clip.encode_text(tok))
            clip.train(was_training)
            self._text = torch.cat(chunks, 0).view(len(self.pathologies), 2, -1)
        return self._text

    @torch.no_grad()
    def predict(self, volume):
        """volume (1, 1, F, H, W) -> (P,) probabilities of 'present' (softmax over each prompt pair)."""
        clip = self.clip
        was_training = clip.training
        clip.eval()
        image_latent = clip.encode_image(volume.to(clip.temperature.device))          # (1, Dl), l2-normalised
        clip.train(was_training)
        assert image_latent.shape[0] == 1, "the reference scores one volume at a time (batch_size=1, zero_shot.py:77-82)"
        logits = (self.text_latents() * image_latent[0]).sum(-1) * clip.temperature.exp()   # (P, 2)
        return torch.softmax(logits, dim=-1)[:, 0]


class CTClipInference(nn.Module):
    """``scripts/zero_shot.py:CTClipInference`` (run_zero_shot.py:36-47).  Additive kwargs: ``dataset`` (anything indexable that
    yields ``(volume, report_text, onehot_labels, accession_name)`` like ``CTReportDatasetinfer``; the reference's NIfTI dataset is
    out of scope and is only built when ``dataset`` is None and the reference's ``data_inference`` module is importable) and
    ``tokenizer``."""

    def __init__(self, CTClip, *, data_folder="external_valid", reports_file="data_reports.xslx", meta_file="meta_data.csv",
                 results_folder="./results", labels="labels.csv", accelerate_kwargs: dict = dict(), dataset=None, tokenizer=None,
                 pathologies=PATHOLOGIES, max_text_len=512):
        super().__init__()
        self.CTClip = CTClip
        if tokenizer is None:
            from transformers import BertTokenizer
            tokenizer = BertTokenizer.from_pretrained("microsoft/BiomedVLP-CXR-BERT-specialized", do_lower_case=True)
        self.tokenizer = tokenizer
        self.register_buffer("steps", torch.Tensor([0]))
        if dataset is None:
            from data_inference import CTReportDatasetinfer   # the reference's scripts/data_inference.py (needs nibabel)
            dataset = CTReportDatasetinfer(data_folder=data_folder, reports_file=reports_file, meta_file=meta_file, labels=labels)
        self.ds = dataset
        self.device = self.CTClip.temperature.device
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(parents=True, exist_ok=True)
        self.classifier = ZeroShotClassifier(CTClip, tokenizer, pathologies, max_length=max_text_len)

    @property
    def is_main(self):
        return _dist.rank() == 0

    def print(self, msg):
        if self.is_main:
            print(msg)

    def infer(self, log_fn=lambda *a, **k: None):
        """Scores every volume of the dataset; writes labels_weights.npz, predicted_weights.npz and accessions.txt like the reference
        (zero_shot.py:153-158).  The AUROC spreadsheet / plots of scripts/eval.py are produced when that module is importable."""
        import numpy as np
        predicted, real, names = [], [], []
        for i in range(len(self.ds)):
            volume, _text, onehot, acc = self.ds[i]
            volume = torch.as_tensor(volume)
            if volume.dim() == 4:
                volume = volume[None]
            predicted.append(self.classifier.predict(volume).float().cpu().numpy())
            real.append(np.asarray(torch.as_tensor(onehot).reshape(-1).cpu()))
            names.append(acc if isinstance(acc, str) else acc[0])
        predicted, real = np.array(predicted), np.array(real)
        out = str(self.results_folder) + "/"
        np.savez(out + "labels_weights.npz", data=real)
        np.savez(out + "predicted_weights.npz", data=predicted)
        with open(out + "accessions.txt", "w") as f:
            f.writelines(n + "\n" for n in names)
        try:
            from eval import evaluate_internal   # the reference's scripts/eval.py (out of scope here)
            evaluate_internal(predicted, real, list(self.classifier.pathologies), out)
        except ImportError:
            pass
        self.steps += 1
        log_fn({})
        self.print("Inference complete")
        return predicted
