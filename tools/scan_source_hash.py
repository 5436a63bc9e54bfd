#!/usr/bin/env python3
"""sha256 of the k_scan region of csrc/index_kernels.hpp (from the "scan: the dominant kernel" banner to the next kernel's
banner: ScanParams, the load / MFMA / emit helpers and k_scan itself).  The committed PMC traffic figure is valid for as
long as THIS text is unchanged; edits to the other kernels of the header do not invalidate it."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan_source_sha256(path=None):
    src = open(path or os.path.join(ROOT, "ano-rag_amd", "csrc", "index_kernels.hpp"), "rb").read()
    a = src.index(b"// scan: the dominant kernel.")
    b = src.index(b"// sample (shadow form of k_scan<DENSE>", a)
    return hashlib.sha256(src[a:b]).hexdigest()


if __name__ == "__main__":
    print(scan_source_sha256(sys.argv[1] if len(sys.argv) > 1 else None))
