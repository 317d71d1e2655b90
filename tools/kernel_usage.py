#!/usr/bin/env python3
"""Register / LDS / scratch use of the projection kernels, from `make -C psa_amd/csrc asm`."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for f in sys.argv[1:] or ["k1_split"]:
    text = (ROOT / "psa_amd/csrc/build" / f"{f}.usage.txt").read_text()
    for block in text.split("Function Name: ")[1:]:
        name = block.split()[0]
        m = re.search(r"k1_split_kernelINS_\d*(\w+?)ELi(\d)ELb(\d)", name)
        tag = f"{m.group(1)} MT{m.group(2)} G{m.group(3)}" if m else name[:48]
        get = lambda key: re.search(key + r": (\d+)", block).group(1)       # noqa: E731
        print(f"{tag:28s} VGPR {get('VGPRs'):>3s} AGPR {get('AGPRs'):>3s} scratch {get('ScratchSize .bytes/lane.'):>4s} "
              f"waves/SIMD {get('Occupancy .waves/SIMD.')} LDS {get('LDS Size .bytes/block.')}")
