#!/usr/bin/env python3
"""One open + decode of a synthetic BAM through the device reader (for rocprofv3).  usage: tools/ingest_prof.py <reads> [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_ingest import synth_bam
from pymasc_amd import bam_device as D
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path = "/tmp/pymasc_ingest_prof_%d.bam" % n
if not os.path.exists(path):
    synth_bam(path, n)
for _ in range(reps):
    t0 = time.time()
    with D.DeviceBamReader(path) as r:
        k = r.decode(10)
        print(k, round(time.time() - t0, 4), r.timings(), flush=True)
