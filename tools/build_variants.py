#!/usr/bin/env python3
"""Build A/B variants of libctmi355.so: python tools/build_variants.py name=-DFLAG=1,-DOTHER=2 name2=..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleantransformer_amd import _build

specs = [a.split("=", 1) for a in sys.argv[1:]]
with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(lambda nv: _build.build_variant(nv[0], [f for f in nv[1].split(",") if f]), specs))
