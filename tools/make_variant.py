#!/usr/bin/env python3
"""A build variant of libmifx.so for tools/ab_gpu.sh:  python tools/make_variant.py NAME [-DFLAG=VALUE ...]  ->  diligentfx_amd/variants/NAME.so
Own object directory per variant (diligentfx_amd/build/variants/NAME), so the shipped library's object cache is left alone; `base` = no flags."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.HERE, "variants")
os.makedirs(out_dir, exist_ok=True)
print(B.build_variant(os.path.join(out_dir, name + ".so"), os.path.join(B.OBJDIR, "variants", name), flags))
