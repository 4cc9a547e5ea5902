#!/usr/bin/env python3
"""Experiment (not a test): can two ranks of the library's RCCL binding share ONE GPU on this box?  RCCL normally refuses
("Duplicate GPU detected"); prints what happens.  tools/rccl_same_gpu_probe.py [world]"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, path, q):
    import numpy as np
    from isonclust2_amd import api
    try:
        ctx = api.Context(0)
        ident = np.zeros(128, np.uint8)
        if rank == 0:
            ctx.L.ioc_dist_unique_id(ident.ctypes.data_as(C.POINTER(C.c_uint8)))
            ident.tofile(path + ".tmp")
            os.rename(path + ".tmp", path)
        else:
            while not os.path.exists(path):
                time.sleep(0.05)
            ident = np.fromfile(path, np.uint8)
        ctx._chk(ctx.L.ioc_dist_init(ctx.h, ident.ctypes.data_as(C.POINTER(C.c_uint8)), rank, world))
        allv = np.zeros(world, np.int64)
        ctx._chk(ctx.L.ioc_dist_allgather_i64(ctx.h, 100 + rank, allv.ctypes.data_as(C.POINTER(C.c_int64))))
        q.put((rank, "ok", allv.tolist()))
        ctx.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", str(e)[:300]))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    path = f"/tmp/ioc_rccl_id_{os.getpid()}"
    c = mp.get_context("spawn")
    q = c.Queue()
    ps = [c.Process(target=worker, args=(r, world, path, q)) for r in range(world)]
    for p in ps:
        p.start()
    try:
        for _ in range(world):
            print(q.get(timeout=90), flush=True)
    except Exception as e:  # noqa: BLE001
        print("no answer:", type(e).__name__, flush=True)
    for p in ps:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
