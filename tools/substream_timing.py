#!/usr/bin/env python3
"""One rank's shard of an N-GPU run rendered as ONE pg_render call versus as K concurrent sub-shards (tiles r + kN of the 2N-way
tiling, one PgScene, HIP stream and host thread each): while one sub-shard's traversal launch drains its tail, the other's kernels fill
the chip.  Timed on ONE GPU for N = 1, 2, 4, 8 (rank 0's tiles); prints one JSON line with ms per frame and the speed-up bound
t(1 GPU, one call) / t(shard) either way.  python tools/substream_timing.py [spp] [K]   GPU box."""
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    import torch
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    pkg = load_package()
    out = {"workload": f"config 3, 1920x1080 @ {spp} spp", "substreams": K, "shards": {}}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "s.pbrt")
        gen_synthetic.write_scene(path, n=708, xres=1920, yres=1080, spp=spp)
        scene = pkg.HostScene(path)
        scenes = [pkg.GpuScene(scene.desc) for _ in range(K)]
        streams = [torch.cuda.Stream() for _ in range(K)]
        dev = torch.device("cuda", 0)

        def bufs(rd):
            n = scenes[0].tile_count(rd)
            ms = n * rd.tile_pixels // 8 + 1024
            return (torch.zeros(max(1, n) * rd.tile_pixels * 4, dtype=torch.float32, device=dev), torch.zeros(ms * 8, dtype=torch.int32, device=dev),
                    torch.zeros(4, dtype=torch.int32, device=dev), ms)

        def frame(jobs):
            """jobs: [(scene, rd, buffers, stream)], one host thread each; returns wall seconds."""
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ts = []
            for gs, rd, b, st in jobs:
                t = threading.Thread(target=lambda gs=gs, rd=rd, b=b, st=st: gs.render_device(rd, b[0].data_ptr(), b[1].data_ptr(), b[3], b[2].data_ptr(), stream=st.cuda_stream))
                t.start(); ts.append(t)
            for t in ts: t.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        t_one = None
        for n in (1, 2, 4, 8):
            rd1 = scene.render_desc(tile_first=0, tile_step=n)
            one = [(scenes[0], rd1, bufs(rd1), streams[0])]
            sub = []
            for k in range(K):
                rdk = scene.render_desc(tile_first=k * n, tile_step=K * n)
                sub.append((scenes[k], rdk, bufs(rdk), streams[k]))
            res = {}
            for name, jobs in (("one_call", one), ("substreams", sub)):
                frame(jobs)
                res[name + "_ms"] = round(min(frame(jobs) for _ in range(3)) * 1e3, 2)
            if n == 1: t_one = res["one_call_ms"]
            res["bound_one_call"] = round(t_one / res["one_call_ms"], 2)
            res["bound_substreams"] = round(t_one / res["substreams_ms"], 2)
            out["shards"][str(n)] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
