"""Pure-torch probe (no libemsanet_hip): a torch reduction captured in a hipGraph vs eager torch
reductions running between capture and replay.  See tools/graph_memset_repro.hip for the HIP-level
twin and DESIGN.md 5b for the finding.   usage: python tools/graph_torch_reduce_repro.py"""
import torch

dev = 'cuda:0'
torch.manual_seed(0)


def nhwc(n, c, h, w, ld=None):
    ld = ld or c
    return torch.randn(n, h, w, ld, device=dev)[..., :c].permute(0, 3, 1, 2)


# the shapes / layouts of the engine's outputs at 96x128 bs 4: channel-sliced NHWC views
xs = [nhwc(4, 40, 96, 128), nhwc(4, 40, 3, 4), nhwc(4, 40, 6, 8), nhwc(4, 40, 12, 16),
      nhwc(4, 1, 96, 128, 8), nhwc(4, 2, 96, 128, 8), nhwc(4, 2, 96, 128, 8),
      nhwc(4, 1, 3, 4, 8), nhwc(4, 2, 3, 4, 8), nhwc(4, 2, 3, 4, 8), torch.randn(4, 16, device=dev)[:, :10]]
big = [torch.randn(1 << 21, device=dev) for _ in range(8)]


def loss_of(ts):
    return sum((t * t).mean() for t in ts)


ref = float(loss_of(xs))
for pre, post in ((0, 0), (1, 0), (0, 1), (1, 1)):
    if pre:
        keep = [o.clone() for o in big]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            loss_of(xs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static = loss_of(xs)
    torch.cuda.synchronize()
    vals = []
    for rep in range(3):
        if post:      # eager multi-block reductions between capture / replays
            junk = [float((o * o).mean()) for o in big] + [float((t * t).mean()) for t in xs]
            same = [torch.equal(o, o) for o in big]
        g.replay()
        torch.cuda.synchronize()
        vals.append(float(static))
    print(f"pre={pre} post={post}: eager {ref:.6f} replays {vals} ->",
          'OK' if all(v == ref for v in vals) else 'MISMATCH')
