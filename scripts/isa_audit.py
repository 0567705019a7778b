"""Two things hipcc does silently that cost this tree real time in round 5, found by reading ISA; this script looks for both in every object of arp_amd/csrc:

  spills   -- registers spilled to scratch per kernel (.vgpr_spill_count of the code object's metadata).  The N = 257 attention instance had 65: hipcc had
              issued all 36 K-fragment LDS reads ahead of the MFMAs and parked twelve of them in scratch on their way (fix: a sched_barrier every six tiles);
              the policy step's masked GEMM instances had 39 after a batch of loads was added to their epilogue (their addresses were formed at the head of
              the kernel and carried across the K loop).
  guarded  -- vector loads followed within four instructions by `s_waitcnt vmcnt(0)`.  A load inside a per-lane `if` (`m < M ? *p : 0`) is a branch around
              the load plus a full wait at the join: the sixteen residual rows per thread that the GEMM epilogues meant to have in flight together were sixteen
              dependent round trips (c_proj -12 %).  Fix: clamp the address, load unconditionally, select the value.  The count includes the scalar tails
              that serve ragged shapes (cold code): read the listing before acting on a number.

  python scripts/isa_audit.py [objects...]        (default: arp_amd/csrc/*.o; needs the built objects, no GPU)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj):
    for f in glob.glob(obj + ".0.*"):
        os.remove(f)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", obj], capture_output=True)
    found = [f for f in glob.glob(obj + ".0.*") if "gfx950" in f]
    return found[0] if found else None


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def audit(obj):
    co = code_object(obj)
    if not co:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    for f in glob.glob(obj + ".0.*"):
        os.remove(f)
    meta = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: int((re.search(rf"\.{k}:\s+(\d+)", blk) or [0, 0])[1])
        meta[name.group(1)] = (g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
    guarded, loads, cur, last = {}, {}, None, -99
    for i, line in enumerate(asm.split("\n")):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur, last = m.group(1), -99
            guarded[cur] = loads[cur] = 0
            continue
        t = line.strip()
        if cur is None:
            continue
        if t.startswith("global_load_dword") or t.startswith("buffer_load"):
            loads[cur] += 1
            last = i
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and i - last <= 4:
            guarded[cur] += 1
            last = -99
    rows = []
    for k, (vg, sp, scr) in meta.items():
        rows.append((os.path.basename(obj), k, vg, sp, scr, guarded.get(k, 0), loads.get(k, 0)))
    return rows


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "arp_amd", "csrc", "*.o")))
    rows = []
    for o in objs:
        rows += audit(o)
    names = demangle(sorted({r[1] for r in rows}))
    print(f"{'object':14s} {'vgpr':>4s} {'spill':>5s} {'scratch B':>9s} {'load->vmcnt(0)':>15s}  kernel")
    for obj, k, vg, sp, scr, gd, ld in sorted(rows, key=lambda r: (-r[3], -r[5])):
        if sp == 0 and gd < 8:
            continue
        print(f"{obj:14s} {vg:4d} {sp:5d} {scr:9d} {gd:7d} of {ld:4d}  {names.get(k, k)[:150]}")


if __name__ == "__main__":
    main()
