#!/usr/bin/env python3
"""Prints the 2^(i/32) table of s3d_expf (sift3d_amd/csrc/s3d_math.h): bits(2^(i/32)) - (i << 47), each entry the
correctly rounded double of a 60-digit decimal evaluation."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60
for i in range(32):
    u = struct.unpack("<Q", struct.pack("<d", float(Decimal(2) ** (Decimal(i) / Decimal(32)))))[0]
    print("0x%016xULL," % (u - (i << 47)), end="\n" if i % 4 == 3 else " ")
