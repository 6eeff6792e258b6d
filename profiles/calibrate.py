#!/usr/bin/env python3
"""Runs the calibration kernels (gfs_test_traffic: known byte counts) -- under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE by
profiles/calibrate.sh, which then divides the counters by the known bytes."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geoflowslam_amd import api
L = api.lib()
cases = [("stream_read_1GiB", 0, 1 << 25, 0, 0),            # 32 Mi records x 32 B, read once
         ("gather32_L2_resident", 1, 1 << 22, 1 << 15, 16),  # 1 MiB table (fits every XCD's L2), 4 Mi threads x 16 gathers = 2 GiB asked for
         ("gather32_hbm", 1, 1 << 22, 1 << 26, 16),          # 2 GiB table: (nearly) every gather misses the caches
         ("stream_write_1GiB", 2, 1 << 25, 0, 0)]
known = {}
for name, mode, n, table, per in cases:
    b = C.c_longlong(0)
    rc = L.gfs_test_traffic(0, mode, n, table, per, C.byref(b))
    assert rc == 0, L.gfs_last_error()
    known[name] = dict(mode=mode, bytes=b.value, kernel=["k_cal_stream_read", "k_cal_gather32", "k_cal_stream_write"][mode], table_bytes=table * 32)
print(json.dumps(known))
