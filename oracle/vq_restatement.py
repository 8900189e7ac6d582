"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ``vector-quantize-pytorch==1.1.2``
(``VectorQuantize(dim, codebook_size, use_cosine_sim=True)``), the third-party dependency the
reference pins at ``transformer_maskgit/setup.py:19`` and calls at ``ctvit.py:18,188,275,403``.

PARITY UNPINNED: the package source is not under /root/reference, is not installed in this image
and has no wheel in the offline wheelhouse, and the reference holds no tests or golden vectors for
it.  What follows restates the published algorithm of lucidrains' 1.1.x ``CosineSimCodebook`` /
``VectorQuantize`` from its documented behaviour (defaults: decay 0.8, eps 1e-5, kmeans_init False,
threshold_ema_dead_code 0, commitment_weight 1.0, sample_codebook_temp 0, no projection because
codebook_dim == dim, heads 1).  State-dict keys: ``_codebook.{initted,cluster_size,embed}``.

Algorithm (train mode), for x of shape (b, n, d):
    flatten = l2norm(x.float());  embed_n = l2norm(embed)
    dist    = flatten @ embed_n^T ;  ind = argmax(dist)           (temperature 0 => plain argmax)
    quant   = embed[ind]                                           (raw, pre-update codebook)
    bins    = histogram(ind);  cluster_size <- lerp(cluster_size, bins, 1-decay)
    esum    = segmented sum of flatten by code; enorm = l2norm(esum / max(bins,1))
    enorm   = where(bins == 0, embed_n, enorm);  embed <- lerp(embed, enorm, 1-decay)
    out     = x + (quant - x).detach()   ; commit loss = mse(quant.detach(), x)  (discarded by ctvit.py:403-412)
Eval mode: out = embed[ind], no buffer updates.
"""
import torch
import torch.nn.functional as F
from torch import nn


def l2norm(t):
    return F.normalize(t, p=2, dim=-1)


class CosineSimCodebook(nn.Module):
    def __init__(self, dim, codebook_size, decay=0.8, eps=1e-5):
        super().__init__()
        self.decay = decay
        self.eps = eps
        self.codebook_size = codebook_size
        embed = torch.empty(1, codebook_size, dim)
        nn.init.kaiming_uniform_(embed)
        embed = l2norm(embed)
        self.register_buffer("initted", torch.Tensor([1.0]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed", embed)

    @torch.no_grad()
    def _ema_update(self, flatten, ind):
        C = self.codebook_size
        onehot = F.one_hot(ind, C).to(flatten.dtype)             # (1, n, C)
        bins = onehot.sum(dim=1)                                  # (1, C)
        self.cluster_size.mul_(self.decay).add_(bins, alpha=1 - self.decay)
        zero_mask = bins == 0
        bins = bins.masked_fill(zero_mask, 1.0)
        embed_sum = torch.einsum("hnd,hnc->hcd", flatten, onehot)
        embed_normalized = l2norm(embed_sum / bins.unsqueeze(-1))
        embed_normalized = torch.where(zero_mask.unsqueeze(-1), l2norm(self.embed), embed_normalized)
        self.embed.mul_(self.decay).add_(embed_normalized, alpha=1 - self.decay)

    def forward(self, x):
        x = x.float()
        shape = x.shape
        flatten = l2norm(x.reshape(1, -1, shape[-1]))
        embed = l2norm(self.embed)
        dist = torch.einsum("hnd,hcd->hnc", flatten, embed)
        ind = dist.argmax(dim=-1)                                 # (1, n)
        quantize = self.embed[0][ind[0]].reshape(shape)           # raw codebook rows, pre-update
        if self.training:
            self._ema_update(flatten, ind)
        return quantize, ind.reshape(shape[:-1])


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, use_cosine_sim=True, decay=0.8, eps=1e-5,
                 commitment_weight=1.0, **unused):
        super().__init__()
        assert use_cosine_sim, "only the cosine-sim codebook (the one ctvit.py:188 builds) is restated"
        self.commitment_weight = commitment_weight
        self.codebook_size = codebook_size
        self._codebook = CosineSimCodebook(dim, codebook_size, decay=decay, eps=eps)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def forward(self, x, mask=None):
        quantize, embed_ind = self._codebook(x)
        if self.training:
            quantize = x + (quantize - x).detach()
        loss = torch.zeros(1, device=x.device, requires_grad=self.training)
        if self.training and self.commitment_weight > 0:
            loss = loss + F.mse_loss(quantize.detach(), x) * self.commitment_weight
        return quantize, embed_ind, loss
