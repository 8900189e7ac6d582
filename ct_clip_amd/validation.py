"""The in-training zero-shot check of the reference trainer (scripts/CTCLIPTrainer.py:266-326: 10 validation volumes x 18 pathologies), kept
for drop-in compatibility of `CTClipTrainer.train()`.  It is the evaluation loop SURVEY.md section 2 marks out of scope: nothing on the
training hot path imports this module (ct_clip_amd/trainer.py calls it only when `evaluate=True`)."""
from pathlib import Path

import torch


def run_validation(self, steps):
    """The in-training zero-shot check of CTCLIPTrainer.py:266-326 (10 validation volumes x 18 pathologies)."""
    import numpy as np
    pathologies = ['Medical material', 'Arterial wall calcification', 'Cardiomegaly', 'Pericardial effusion',
                   'Coronary artery wall calcification', 'Hiatal hernia', 'Lymphadenopathy', 'Emphysema', 'Atelectasis',
                   'Lung nodule', 'Lung opacity', 'Pulmonary fibrotic sequela', 'Pleural effusion',
                   'Mosaic attenuation pattern', 'Peribronchial thickening', 'Consolidation', 'Bronchiectasis',
                   'Interlobular septal thickening']
    model = self.CTClip
    model.eval()
    predictedall, realall = [], []
    with torch.no_grad():
        for _ in range(10):
            valid_data, text, onehotlabels, name_acc = next(self.valid_dl_iter)
            valid_data = valid_data.to(self.device)
            predicted = []
            for pathology in pathologies:
                tokens = self.tokenize([f"There is {pathology}.", f"There is no {pathology}."])
                out = torch.softmax(model(tokens, valid_data, device=self.device), dim=0)
                predicted.append(float(out[0]))
            predictedall.append(predicted)
            realall.append(onehotlabels.detach().cpu().numpy()[0])
    model.train()
    plotdir = str(self.results_folder / f"CTClip_{steps}") + "/"
    Path(plotdir).mkdir(parents=True, exist_ok=True)
    try:
        from eval import evaluate_internal  # the reference's scripts/eval.py
        evaluate_internal(np.array(predictedall), np.array(realall), pathologies, plotdir)
    except ImportError:
        np.save(plotdir + "predicted.npy", np.array(predictedall))
        np.save(plotdir + "labels.npy", np.array(realall))
