"""Instruction mix of the loops of a csrc/*.hip file's kernels, from the compiler's assembly (no GPU needed).

On gfx950 fp32 MFMA and vector instructions of one SIMD do not overlap (tools/mfma_probe.hip), so every vector
instruction inside a tile loop is paid in matrix-pipe time; the quarter-rate integer multiplies most of all.  This lists,
per kernel: registers, occupancy, and for the innermost loop the MFMA count, the other vector instructions and the slow
ones (v_mul_lo / v_mul_hi / v_mad_u64) -- all blocks of the loop, so conditional paths (last tile, key hole, lazy
rescale) are included.
    python tools/isa_loop_mix.py [file.s] [--src loss.hip] [-v] [--waits]
(without a .s file: compiles csrc/<src>, default attn.hip).  --waits prints, per kernel, the order of the loop's vector
memory instructions and `s_waitcnt vmcnt(N)`: L = load, S = store, wN = wait until at most N are outstanding.  The
counter is ONE for loads and stores, in issue order; behind a branch that only some path takes the compiler no longer
knows N and waits for more than the data it needs -- `S w0` ahead of arithmetic means "wait for that store to be
acknowledged", `L L L w0` right after a prefetch means the prefetch is not one.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(src="attn.hip"):
    out = os.path.join(tempfile.mkdtemp(), src.replace(".hip", ".s"))
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "vqattack_amd", "csrc", src)], check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    """Per kernel of an assembly file: (demangled name, {NumVgprs, Occupancy, ScratchSize}, loop body text)."""
    text = open(path).read()
    for k in re.split(r"\n(?=_ZN3vqa[^\n]*:\s+; @)", text):
        name = k.split(":")[0]
        if not name.startswith("_ZN3vqa"):
            continue
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].strip()
        demangled = demangled[5:] if demangled.startswith("void ") else demangled
        meta = {key: (re.search(r"; %s: (\d+)" % key, k) or [None, "?"])[1] for key in ("NumVgprs", "Occupancy", "ScratchSize")}
        # basic blocks start at a label or a "; %bb.N:" comment; those of a loop say "Loop Header" / "in Loop:"
        body, inside = [], False
        for line in k.split("\n"):
            if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
                inside = "Loop Header" in line or "in Loop:" in line
            elif inside:
                body.append(line)
        yield demangled, meta, "\n".join(body)


def memory_sequence(body):
    """The loop's vector memory instructions and vmcnt waits in layout order: L, S, A(tomic), wN, | (barrier)."""
    seq = []
    for line in body.split("\n"):
        t = line.split()
        if not t or not line.startswith("\t"):
            continue
        if t[0].startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif t[0].startswith(("global_store", "buffer_store")):
            seq.append("S")
        elif t[0].startswith(("global_atomic", "buffer_atomic")):
            seq.append("A")
        elif t[0] == "s_waitcnt" and "vmcnt" in line:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", line).group(1))
        elif t[0] == "s_barrier":
            seq.append("|")
    return seq


def main():
    args = sys.argv[1:]
    src = args[args.index("--src") + 1] if "--src" in args else "attn.hip"
    files = [a for a in args if a.endswith(".s")]
    path = files[0] if files else assembly(src)
    verbose, waits = "-v" in args, "--waits" in args
    for demangled, meta, body in kernels(path):
        ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.split() and not l.strip().startswith(";")]
        valu = [x for x in ins if x.startswith("v_") and not x.startswith("v_mfma")]
        slow = [x for x in valu if x.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64"))]
        print("{:<62} vgpr {:>3} occ {} scratch {} | loop: mfma {:>3} valu {:>3} (slow {:>2}) lds {:>3} vmem {:>2}".format(
            demangled, meta["NumVgprs"], meta["Occupancy"], meta["ScratchSize"],
            sum(x.startswith("v_mfma") for x in ins), len(valu), len(slow), sum(x.startswith("ds_") for x in ins),
            sum(x.startswith(("global_", "buffer_")) for x in ins)))
        if verbose and body:
            print("    " + ", ".join("{} {}".format(n, c) for n, c in collections.Counter(valu).most_common()))
        if waits and body:
            print("    " + " ".join(memory_sequence(body)))


if __name__ == "__main__":
    main()
