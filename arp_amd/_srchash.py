"""sha1 over the kernel sources (arp_amd/csrc/*.{h,hip,cpp}, Makefile): profiles/pmc_traffic*.json are stamped with it by scripts/summarize_*.py and
bench.py prints `traffic_stale: true` when the tree's differs -- a committed counter file must not outlive the kernels it measured."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def csrc_sha1():
    h = hashlib.sha1()
    files = sorted(f for pat in ("*.h", "*.hip", "*.cpp", "Makefile") for f in glob.glob(os.path.join(_HERE, "csrc", pat)))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha1())
