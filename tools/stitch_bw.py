import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
if len(sys.argv) > 1:
    from envpool_amd.dummy.dummy_envpool import HostBlocks, stitch_parts
    rows, K = 4096, 8
    n = rows // K
    shapes = [((160, 41), np.uint8), ((23,), np.uint8), ((16, 12), np.uint8), ((32, 14), np.uint8), ((), np.int32), ((16,), np.uint8)]
    src_kind = sys.argv[1]
    hb_src = HostBlocks(64 << 20)
    def mk(shape, dt):
        if src_kind == "pinned":
            b = hb_src.take()
            return b.array(0, (n,) + shape, np.dtype(dt).str)
        return np.ones((n,) + shape, dt)
    parts = [[mk(s, d) for s, d in shapes] for _ in range(K)]
    offs, at = [], 0
    for s, d in shapes:
        offs.append(at)
        at = (at + np.dtype(d).itemsize * int(np.prod(s, dtype=np.int64)) * rows + 255) // 256 * 256
    hb = HostBlocks(at)
    total = sum(a.nbytes for p in parts for a in p)
    ts = []
    for i in range(40):
        blk = hb.take()
        t = time.perf_counter(); stitch_parts(blk, offs, parts); ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"src={src_kind} threads={os.environ.get('EP_STITCH_THREADS')}: {total/1e6:.1f} MB in {ts[len(ts)//2]*1e6:.0f} us = {total/ts[len(ts)//2]/1e9:.1f} GB/s", flush=True)
    t = time.perf_counter()
    for i in range(10):
        out = [np.concatenate([p[k] for p in parts]) for k in range(len(shapes))]
    print(f"   np.concatenate: {(time.perf_counter()-t)/10*1e6:.0f} us", flush=True)
else:
    for kind in ("numpy", "pinned"):
        for th in (1, 2, 4, 6, 8, 12, 16):
            subprocess.run([sys.executable, __file__, kind], env=dict(os.environ, EP_STITCH_THREADS=str(th)), check=True)
