#!/usr/bin/env python3
"""Static check of the ring form of the fused encoder-head forward kernel (csrc/enc12_tile.hpp, RING = 1, camera bytes): its frame loads are inline assembly and
their s_waitcnt is written by hand, so nothing protects a register between `global_load_dword` and the wait except this check.  Reads the hipcc -S listing of
enc12.hip (compiled here if no path is given), walks the kernel in layout order -- once from the top, then the band loop a second time for the loop-carried
requests -- with the hardware's rule (loads retire in order; a wait vmcnt(N) leaves at most N of them outstanding; stores only ever make a wait stricter) and fails if any
instruction READS a register whose load may still be outstanding, or if the kernel spills.
    python tools/check_enc12_isa.py [listing.s]
    python tools/check_enc12_isa.py --generic carla-ppo_amd/csrc/ares.hip "ares_(conv|gather|gather2)_kernel"      (any file / kernels: the replay alone)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["_ZN2mi16enc12_fwd_kernelIhLi0ELi1ELi0E", "_ZN2mi16enc12_fwd_kernelIhLi0ELi1ELi1E"]      # ring form; ring form + pipelined conv2 reads


def listing(kernel, path=None):
    KERNEL = kernel
    if path is None:
        path = os.path.join(tempfile.mkdtemp(), "enc12.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(ROOT, "carla-ppo_amd", "csrc", "enc12.hip"), "-o", path], check=True, capture_output=True)
    out, on = [], False
    for ln in open(path):
        if ln.startswith(KERNEL) and ":" in ln.split(";")[0]:
            on = True
        if on:
            out.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    meta = open(path).read()
    return out, meta, path


def regs_of(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []


def walk(lines, outstanding, errors, tag, lds=None):
    lds = [] if lds is None else lds                       # outstanding LGKM operations in issue order: the registers an LDS read will write ([] for writes / scalar loads)
    for no, ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op, _, rest = t.partition(" ")
        ops = [x for x in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                del outstanding[:max(0, len(outstanding) - int(m.group(1)))]
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                del lds[:max(0, len(lds) - int(m.group(1)))]
            continue
        if op.startswith("ds_read"):
            for tok in ops[1:]:
                for r in regs_of(tok.strip().split(" ")[0]):
                    if r in outstanding or any(r in g for g in lds):
                        errors.append("%s line %d: address v%d of an LDS read is itself outstanding" % (tag, no, r))
            lds.append(regs_of(ops[0]))
            continue
        if op.startswith(("ds_write", "s_load")):
            lds.append([])
        if op == "global_load_dword" and len(ops) >= 2 and ops[-1].strip() == "off":      # the ring's loads (the compiler's own loads carry an SGPR base or a wider type)
            for r in regs_of(ops[1]) + (regs_of(ops[2]) if len(ops) > 3 else []):
                if r in outstanding:
                    errors.append("%s line %d: address register v%d of a load is itself an outstanding load" % (tag, no, r))
            outstanding.extend(regs_of(ops[0]))
            continue
        reads = ops[1:] if (op.startswith(("v_", "ds_read", "global_load", "buffer_load")) and not op.startswith("v_cmpx")) else ops
        if op.startswith(("global_store", "ds_write", "buffer_store", "scratch_store")):
            reads = ops
        for tok in reads:
            for r in regs_of(tok.strip().split(" ")[0]):
                if r in outstanding or any(r in g for g in lds):
                    errors.append("%s line %d: `%s` reads v%d while its load may be outstanding" % (tag, no, t, r))


def check(kernel, path):
    lines, meta, path = listing(kernel, path)
    if not lines:
        raise SystemExit("kernel %s not found in the listing" % kernel)
    num = list(enumerate(lines, 1))
    errors, outstanding, lds = [], [], []
    walk(num, outstanding, errors, "pass 1", lds)
    # the band loop: from the first loop header on (layout order), walked again with what pass 1 left outstanding
    heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    if heads:
        walk(num[heads[0] - 1:], outstanding, errors, "pass 2 (band loop)", lds)
    m = re.search(re.escape(kernel) + r"[^\n]*\n(?:.*\n){0,80}?\s*\.vgpr_spill_count:\s*(\d+)", meta)
    spills = int(m.group(1)) if m else -1
    n_loads = sum(1 for l in lines if re.match(r"\s*global_load_dword v\d+, v\[\d+:\d+\], off", l))
    n_waits = sum(1 for l in lines if re.search(r"s_waitcnt vmcnt\(12\)", l))
    n_lds = sum(1 for l in lines if re.match(r"\s*ds_read_b128 v\[\d+:\d+\], v\d+ offset:", l) or re.match(r"\s*ds_read_b128 v\[\d+:\d+\], v\d+\s*$", l))
    print("kernel %s: %d ring loads, %d hand-written vmcnt waits, %d LDS fragment reads, vgpr spills %d, %d violations" % (kernel[22:], n_loads, n_waits, n_lds, spills, len(errors)))
    for e in errors[:20]:
        print("  " + e)
    return path, bool(errors) or spills != 0 or n_loads < 36 or n_waits < 3


def check_generic(hip_path, pattern, listing_path=None):
    """Any kernel of `hip_path` whose mangled name matches `pattern`: the same replay (hand-written LDS reads / waits of the ar_lds_read idiom of ares_tile.hpp and rwconv.hip are covered by
    the LGKM half of the rule), no expectation about counts.  Returns True when a violation was found."""
    if listing_path is None:
        listing_path = os.path.join(tempfile.mkdtemp(), "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        hip_path, "-o", listing_path], check=True, capture_output=True)
    txt = open(listing_path).read().split("\n")
    starts = {}
    for i, ln in enumerate(txt):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", ln)
        if m and re.search(pattern, m.group(1)):
            starts[m.group(1)] = i
    bad, n_asm = False, 0
    for k, i0 in starts.items():
        out = []
        for ln in txt[i0:]:
            out.append(ln)
            if "s_endpgm" in ln:
                break
        num = list(enumerate(out, 1))
        errors, outstanding, lds = [], [], []
        walk(num, outstanding, errors, "pass 1", lds)
        heads = [i for i, l in enumerate(out) if "Loop Header: Depth=1" in l]
        if heads:
            walk(num[heads[0] - 1:], outstanding, errors, "pass 2", lds)
        n_asm += sum(1 for l in out if "ASMSTART" in l)
        if errors:
            bad = True
            print("kernel %s: %d violations" % (k[:80], len(errors)))
            for e in errors[:5]:
                print("  " + e)
    print("%s: %d kernels matching /%s/, %d inline-assembly statements, %s" % (os.path.basename(hip_path), len(starts), pattern, n_asm, "VIOLATIONS" if bad else "0 violations"))
    return bad or not starts


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--generic":          # --generic file.hip name-regex [listing.s]
        raise SystemExit(1 if check_generic(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None) else 0)
    path, bad = sys.argv[1] if len(sys.argv) > 1 else None, False
    for k in KERNELS:
        path, b = check(k, path)
        bad = bad or b
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
