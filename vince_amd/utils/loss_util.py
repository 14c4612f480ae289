"""similarity_cross_entropy with the reference's signature (utils/loss_util.py:7-62) on the HIP row kernel.

The hot path never comes here -- VinceModel.forward fuses similarity + loss (csrc/infonce.hip) -- but the function is
part of the reference's public surface (VinceModel.loss calls it, end tasks may too), so it exists for callers that
hold a materialised similarity matrix.  Only the equal-positives-per-row path is implemented; the reference's
USE_FLOAT branch (unequal counts, a process-wide cached decision, App. D item 2) raises.
"""
import ctypes

import torch

from .. import ops
from .._lib import check, lib


def _mask_from_spec(spec, device):
    """VinceModel.forward hands out masks as compact specs; expand one to a boolean matrix on demand."""
    kind, frames, b, k = spec
    if kind == "first_column":
        m = torch.zeros(b, k + 1, dtype=torch.bool, device=device)
        m[:, 0] = True
        return m
    r = torch.arange(b, device=device)
    m = torch.zeros(b, b + k, dtype=torch.bool, device=device)
    m[:, :b] = (r[:, None] // frames) == (r[None, :] // frames)
    return m


class _SceRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sims, mask_u8, p, inv_t):
        b, cols = sims.shape
        dists = torch.empty(b, p, device=sims.device)
        sw = torch.empty(b, p, device=sims.device)
        rmax = torch.empty(b, device=sims.device)
        nsum = torch.empty(b, device=sims.device)
        sc = sims.detach().contiguous()
        check(lib().vince_sce_rows_fwd(ops._ptr(sc), ops._ptr(mask_u8), b, cols, p, inv_t, ops._ptr(dists), ops._ptr(sw),
                                       ops._ptr(rmax), ops._ptr(nsum), ops.stream_ptr()))
        ctx.saved = (sc, mask_u8, p, inv_t, rmax, nsum)
        ctx.mark_non_differentiable(sw)
        return dists, sw

    @staticmethod
    def backward(ctx, g_dists, _g_sw):
        sc, mask_u8, p, inv_t, rmax, nsum = ctx.saved
        b, cols = sc.shape
        dsims = torch.empty_like(sc)
        gd = g_dists.contiguous().float()
        check(lib().vince_sce_rows_bwd(ops._ptr(sc), ops._ptr(mask_u8), b, cols, p, inv_t, ops._ptr(rmax), ops._ptr(nsum),
                                       ops._ptr(gd), ops._ptr(dsims), ops.stream_ptr()))
        return dsims, None, None, None


def similarity_cross_entropy(similarities, temperature, n_feat, n_rows1, mask=None, n_positives_per_row=None):
    if hasattr(similarities, "materialize"):
        similarities = similarities.materialize()
    ops.require_gpu(similarities.detach() if similarities.is_contiguous() else similarities.detach().contiguous())
    if mask is None:
        assert n_positives_per_row is not None
        mask = (torch.eye(n_feat, device=similarities.device, dtype=torch.bool)
                .repeat_interleave(n_positives_per_row, 1).repeat_interleave(n_rows1, 0))
    elif isinstance(mask, tuple):
        mask = _mask_from_spec(mask, similarities.device)
    assert mask.shape == similarities.shape
    counts = mask.sum(-1)
    p = int(counts[0])
    if not bool((counts == p).all()):
        raise NotImplementedError("similarity_cross_entropy: rows with different numbers of positives (the reference's "
                                  "USE_FLOAT branch) are not implemented on the HIP path")
    b = similarities.shape[0]
    dists, sw = _SceRowsFn.apply(similarities.float(), mask.to(torch.uint8).contiguous(), p, 1.0 / temperature)
    dists = dists.view(n_feat, n_rows1, p)
    sw = sw.view(n_feat, n_rows1, p)
    return dict(dists=dists, dist=dists.mean(), softmax_weights=sw, softmax_weight=sw.mean())
