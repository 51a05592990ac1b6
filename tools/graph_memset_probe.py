"""Does a hipMemsetAsync node captured into a hipGraph keep its stream-order edge to the kernel behind it?  (the round-1
rasteriser fault was attributed to the z-buffer memset overlapping the resolve pass on graph replay; this probe checks that claim in
isolation.)  Captures  memset(buf, 0xFF) -> copy kernel(dst <- buf)  on one stream, dumps the graph as DOT, and replays it after
overwriting `buf` with another pattern each time: a replay whose `dst` is not all 0xFF ran the kernel before / across the memset.
usage: python tools/graph_memset_probe.py [out.dot]"""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402
from lib.hip import ops  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "graph_memset_probe.dot"
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
n = 16 * 480 * 640 * 2   # words: the z-buffer of a 16-pair batch (39 MB)
buf = torch.zeros(n, dtype=torch.int32, device="cuda:0")
dst = torch.zeros(n, dtype=torch.int32, device="cuda:0")


def body():
    rc = hip.hipMemsetAsync(buf.data_ptr(), 0xFF, n * 4, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    ops.copy(dst, buf)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    body()
g.debug_dump(os.path.abspath(out))
if os.path.exists(out):
    dot = open(out).read()
    nodes = re.findall(r'^\s*"?(\w+)"?\s*\[.*?label="([^"]*)"', dot, flags=re.M)
    edges = re.findall(r'^\s*"?(\w+)"?\s*->\s*"?(\w+)"?', dot, flags=re.M)
    print("nodes:", [(a, b.split("\\n")[0][:40]) for a, b in nodes])
    print("edges:", edges)
else:
    print("hipGraphDebugDotPrint wrote no file on this runtime: replay check only")
bad = 0
for rep in range(50):
    buf.fill_(rep + 1)
    dst.fill_(-7)
    g.replay()
    torch.cuda.synchronize()
    if not bool((dst == -1).all()):
        bad += 1
print("replays with a stale / torn copy: {} of 50".format(bad))
