"""k-nearest-neighbour evaluation of embeddings on the GPU.

The reference scores its encoder on CIFAR with an sklearn KDTree: 11 Euclidean neighbours of every embedding, the self
match dropped, the most frequent label of the remaining 10 as the prediction (solvers/vince_solver.py:651-679).  Here
the neighbours come from a brute-force distance matrix: block of queries x whole bank through the fp32 MFMA GEMM
(`ops.linear_fwd`, i.e. vince_conv_igemm) as  2 q.b - |b|^2  (= |q|^2 - d^2: the same ordering as the Euclidean
distance, exact fp32), then a top-k per row.  No CPU path: tensors must be on the GPU.
"""
import torch

from .. import ops


def knn_indices(features, k, block_rows=4096):
    """Indices [N, k] of each row's k nearest rows of `features` [N, D] (fp32, GPU) by Euclidean distance, nearest first;
    the row itself is normally column 0 (distance 0), exactly as `KDTree.query(x, k)` returns it."""
    ops.require_gpu(features)
    if features.dim() == 4:     # vince_solver.py:668-670: spatial features are averaged first
        features = features.mean(dim=(2, 3))
    feats = features.float().contiguous()
    n, d = feats.shape
    if k > n:
        raise ValueError("knn_indices: k=%d exceeds the %d rows" % (k, n))
    pad_n = (-n) % 4           # the GEMM writes 16-byte chunks of output channels
    pad_d = (-d) % 4
    bank = torch.nn.functional.pad(feats, (0, pad_d, 0, pad_n))
    norms = (bank * bank).sum(1)
    if pad_n:
        norms[n:] = float("inf")   # padding rows can never be neighbours
    weight = (2.0 * bank).contiguous()
    bias = (-norms).contiguous()
    out = torch.empty(n, k, dtype=torch.int64, device=feats.device)
    for r0 in range(0, n, block_rows):
        q = bank[r0:min(n, r0 + block_rows)].contiguous()
        score = ops.linear_fwd(q, weight, bias)          # [rows, N + pad] = 2 q.b - |b|^2
        out[r0:r0 + q.shape[0]] = torch.topk(score, k, dim=1).indices
    return out


def knn_accuracy(features, labels, k=10):
    """Leave-one-out k-NN accuracy as the reference computes it: k+1 neighbours, first one (the self match) dropped,
    prediction = the most frequent label among the rest, ties -> the smallest label (scipy.stats.mode)."""
    ops.require_gpu(features, labels)
    labels = labels.to(features.device).long().view(-1)
    nbrs = knn_indices(features, k + 1)[:, 1:]
    votes = labels[nbrs]                                                # [N, k]
    n_cls = int(labels.max()) + 1
    counts = torch.zeros(votes.shape[0], n_cls, dtype=torch.int32, device=votes.device)
    counts.scatter_add_(1, votes, torch.ones_like(votes, dtype=torch.int32))
    preds = counts.argmax(dim=1)                                        # first maximum = smallest label
    return float((preds == labels).float().mean()), preds, nbrs
