#!/usr/bin/env python3
"""Does reserving a few CUs for the K_uu chain pay at the 8-GPU shard size?  Pass 1 + tail run on a stream whose CU mask
excludes k CUs (spread over the XCDs), the side chain on a stream masked to exactly those k; k = 0 is the product path.
    python3 tools/cu_mask_probe.py [rows ...]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(cus, ncu=256):
    words = (ncu + 31) // 32
    mask = [0] * words
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    arr = (C.c_uint32 * words)(*mask)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    rows_list = [int(v) for v in sys.argv[1:]] or [125_000, 250_000, 1_000_000]
    eng = ggp_amd.HipEngine()
    ncu = torch.cuda.get_device_properties(eng.device).multi_processor_count
    for rows in rows_list:
        X, y, Z = bench.synth(rows, bench.M_IND, bench.DIM)
        Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
        for k, stride_desc in ((0, "-"), (8, "1 per XCD"), (16, "2 per XCD"), (32, "4 per XCD")):
            cb = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng)
            if k:
                # mask bit i is CU (i // 8) of XCD (i % 8) on this part (probed: bits 0, 32, 64 ... all landed in one XCD and
                # slowed the contraction by 32 / 24): the first k bits are k / 8 CUs in each of the 8 XCDs
                reserved = list(range(k))
                side = masked_stream(reserved, ncu)
                main = masked_stream([c for c in range(ncu) if c not in set(reserved)], ncu)
                cb._side, cb._pool = side, None
                # the helper thread announces its CU budget once
                from concurrent.futures import ThreadPoolExecutor
                pool = ThreadPoolExecutor(max_workers=1)
                pool.submit(lambda: eng.lib.sgp_set_cu_budget(k)).result()
                cb._pool = pool
                eng.lib.sgp_set_cu_budget(ncu - k)
            else:
                main = torch.cuda.current_stream(eng.device)
                eng.lib.sgp_set_cu_budget(0)
            fn = lambda: cb.value(Zd, [bench.LS] * bench.DIM, bench.SF ** 2, bench.SN ** 2)
            with torch.cuda.stream(main):
                for _ in range(5):
                    F = fn()[0]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                K = 30
                for _ in range(K):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / K * 1e3
            print(json.dumps({"rows": rows, "reserved_cus": k, "layout": stride_desc, "ms_per_eval": round(dt, 3), "F": F}), flush=True)
        eng.lib.sgp_set_cu_budget(0)


if __name__ == "__main__":
    main()
