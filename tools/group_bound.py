"""What would ONE grouped launch per layer for both encoders buy (VERDICT r4 next #1e)?  An upper bound, measured without building it:
a grouped launch has twice the tiles / rows of today's per-encoder launch and the same fixed cost, which is exactly what ONE encoder's
forward at batch 2B looks like to the machine (same kernels, same layer sequence, twice the work per launch).  Compared here, no-grad
train-mode ResNet-50 forwards, bf16:
    A   two forwards at B back to back on one stream          (no overlap, today's launches)
    B   two forwards at B on two streams                      (today's arrangement: key encoder beside the query encoder)
    C   one forward at 2B                                     (the launch structure of grouped key + query launches)
B - C is what grouping can still win over the two-stream overlap.  Usage: python tools/group_bound.py [B=256] [reps=10]"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd.config import make_args
from vince_amd.models.vince_model import VinceModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def model(batch):
    args = make_args(backbone="ResNet50", vince_embedding_size=128, compute_dtype="bf16", batch_size=batch, input_size=(224, 224))
    m = VinceModel(args).to("cuda:0")
    m.train()
    m.clone_spatial = False
    return m


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    q, k = model(B), model(B)
    x1, x2 = torch.randn(B, 3, 224, 224, device="cuda:0"), torch.randn(B, 3, 224, 224, device="cuda:0")
    side = torch.cuda.Stream()

    def seq():
        q.get_embeddings({"data": x1})
        k.get_embeddings({"data": x2})

    def two_streams():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            k.get_embeddings({"data": x2})
        q.get_embeddings({"data": x1})
        main.wait_stream(side)

    a = timed(seq)
    b = timed(two_streams)


def two_streams_grad():
    # D: the training step's own pair -- the query encoder GRAD-ENABLED (saves what backward reads) beside the no-grad key encoder
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        k.get_embeddings({"data": x2})
    q.get_embeddings({"data": x1})
    main.wait_stream(side)


def one_grad():
    q.get_embeddings({"data": x1})


d = timed(two_streams_grad)
d1 = timed(one_grad)
print("D grad-enabled query + no-grad key on two streams %.3f ms (grad-enabled forward alone %.3f ms)" % (d, d1))
with torch.no_grad():
    del k
    torch.cuda.empty_cache()
    big = model(2 * B)
    xx = torch.cat([x1, x2])
    c = timed(lambda: big.get_embeddings({"data": xx}))
    one = timed(lambda: q.get_embeddings({"data": x1}))
print("B=%d: one forward %.3f ms | A two forwards, one stream %.3f ms | B two forwards, two streams %.3f ms | C one forward at 2B %.3f ms | "
      "grouping bound B - C = %.3f ms (A - C = %.3f)" % (B, one, a, b, c, b - c, a - c))
