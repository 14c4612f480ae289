"""similarity_cross_entropy with the reference's signature (utils/loss_util.py:7-62) on the HIP row kernel.

The hot path never comes here -- VinceModel.forward fuses similarity + loss (csrc/infonce.hip) -- but the function is
part of the reference's public surface (VinceModel.loss calls it, end tasks may too), so it exists for callers that
hold a materialised similarity matrix.  Both paths of the reference are here: equal positives per row (positives compacted
to [n_feat, n_rows1, P]) and the USE_FLOAT branch for unequal counts (loss_util.py:30-35,47-49: full-width outputs, the
-2**20 fill carried through, means over the mask entries) -- including the reference's process-wide cache of that decision
(loss_util.py:4,28-29, App. D item 2): the FIRST mask this process sees fixes the path for every later call.
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
    """Returns (dists, softmax weights, neg_sum); neg_sum -- the row kernel's exp-sum over the non-mask columns, relative to the
    row maximum -- is an extra non-differentiable output (the USE_FLOAT fill values), so every call owns its own."""

    @staticmethod
    def forward(ctx, sims, mask_u8, p, inv_t):
        b, cols = sims.shape
        dists = torch.zeros(b, p, device=sims.device)     # (rows with fewer than p positives leave their tail at zero)
        sw = torch.zeros(b, p, device=sims.device)
        rmax = torch.empty(b, device=sims.device)
        nsum = torch.empty(b, device=sims.device)
        sc = sims.detach().contiguous()
        check(lib().vince_sce_rows_fwd(ops._ptr(sc), ops._ptr(mask_u8), b, cols, p, inv_t, ops._ptr(dists), ops._ptr(sw),
                                       ops._ptr(rmax), ops._ptr(nsum), ops.stream_ptr()))
        ctx.saved = (sc, mask_u8, p, inv_t, rmax, nsum)
        ns_out = nsum.clone()
        ctx.mark_non_differentiable(sw, ns_out)
        return dists, sw, ns_out

    @staticmethod
    def backward(ctx, g_dists, _g_sw, _g_ns):
        sc, mask_u8, p, inv_t, rmax, nsum = ctx.saved
        b, cols = sc.shape
        dsims = torch.empty_like(sc)
        gd = g_dists.contiguous().float()
        check(lib().vince_sce_rows_bwd(ops._ptr(sc), ops._ptr(mask_u8), b, cols, p, inv_t, ops._ptr(rmax), ops._ptr(nsum),
                                       ops._ptr(gd), ops._ptr(dsims), ops.stream_ptr()))
        return dsims, None, None, None


USE_FLOAT = None   # loss_util.py:4 -- decided by the first call, then kept (reset to None to decide again, as the reference allows)


def similarity_cross_entropy(similarities, temperature, n_feat, n_rows1, mask=None, n_positives_per_row=None):
    global USE_FLOAT
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
    if USE_FLOAT is None:
        USE_FLOAT = bool(counts.min() != counts.max())          # loss_util.py:28-29
    p = int(counts.max())
    if not USE_FLOAT and not bool((counts == p).all()):
        # the reference's .view(n_feat, n_rows1, -1) of the compacted positives fails here (loss_util.py:38-39)
        raise RuntimeError("similarity_cross_entropy: rows with different numbers of positives after an equal-count mask fixed "
                           "USE_FLOAT = False for this process (reference loss_util.py:28-29,38-39); set loss_util.USE_FLOAT = None")
    b, cols = similarities.shape
    # one row kernel for both paths: positives compacted per row to [b, p] (rows with fewer positives leave zeros at the end)
    dists_c, sw_c, neg_sum = _SceRowsFn.apply(similarities.float(), mask.to(torch.uint8).contiguous(), p, 1.0 / temperature)
    if not USE_FLOAT:
        dists = dists_c.view(n_feat, n_rows1, p)
        sw = sw_c.view(n_feat, n_rows1, p)
        return dict(dists=dists, dist=dists.mean(), softmax_weights=sw, softmax_weight=sw.mean())
    # USE_FLOAT: scatter the compacted values back to their columns; everywhere else the reference's fill arithmetic gives
    # dists = -(-2**20 - log(neg_sum)) in fp32 and weights = exp(-2**20 - ...) = 0
    rank = torch.cumsum(mask.to(torch.int64), dim=1) - 1                       # position of each positive inside its row
    rows = torch.arange(b, device=mask.device).unsqueeze(1).expand_as(mask)
    fill = -(torch.full((b, 1), -2.0 ** 20, device=mask.device) - torch.log(neg_sum.view(b, 1)))
    dists = fill.expand(b, cols).clone()
    sw = torch.zeros(b, cols, device=mask.device)
    dists[mask] = dists_c[rows[mask], rank[mask]]
    sw[mask] = sw_c[rows[mask], rank[mask]]
    dists = dists.view(n_feat, n_rows1, cols)
    sw = sw.view(n_feat, n_rows1, cols)
    m3 = mask.view(n_feat, n_rows1, cols)
    return dict(dists=dists, dist=dists[m3].mean(), softmax_weights=sw, softmax_weight=sw[m3].mean())
