#!/usr/bin/env python3
"""One linearized pass of the C3 scene (for rocprofv3 --stats): python tools/c3_lin_run.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402
import bench_secondary as BS  # noqa: E402

arch = vsm.Architectures.GPU(0)
e = BS.c3_lin(vsm, torch, arch)
print(e)
