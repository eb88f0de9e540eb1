#!/usr/bin/env python3
"""Audit of conv_halo2_kernel's ISA (cdna_hip_programming.md section 5.7, item 1: "an asm load's VGPR destination counts as written
at ;;#ASMEND, so the compiler may read, copy, spill or reuse it before the data lands").

The kernel's register loads (buffer_load_dwordx4 of weight fragments, ds_read_b128 of pixel fragments) are inline asm with hand-counted
waits.  This script walks the K loop of every instantiation in a `hipcc -S` listing, simulates the two queues -- VMEM (every
buffer / global / scratch load incl. the LDS-DMA pieces, retired in order by `s_waitcnt vmcnt(N)`) and LGKM (ds_read, retired by
`lgkmcnt(0)`) -- and reports every instruction that READS or WRITES a register whose load has not been waited for.  Exit code 1 on
a finding.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/conv_halo2_bf16.hip -o /tmp/h2.s
  python scripts/h2_audit.py /tmp/h2.s"""
import re, sys

def regs(tok):
    """v[a:b] / v5 / a[..] -> set of ('v'|'a', index)"""
    out = set()
    for kind, a, b in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(kind, i) for i in range(int(a), int(b) + 1)}
    for kind, a in re.findall(r"(?<![\w\[])([va])(\d+)\b", tok):
        out.add((kind, int(a)))
    return out

def audit(lines, name):
    vm = []            # pending VMEM ops in issue order: set of dest regs (empty for LDS-DMA / stores)
    lg = []            # pending ds_reads
    findings = []
    in_loop = False
    for ln, raw in lines:
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        args = s[len(op):]
        pending = set().union(*[x for x in vm if x != "LDSDMA"]) if vm else set()
        pending |= set().union(*lg) if lg else set()
        touched = regs(args)
        if op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                vm = vm[len(vm) - n:] if n < len(vm) else vm
                if n == 0: vm = []
            if re.search(r"lgkmcnt\(0\)", s):
                lg = []
            continue
        is_vm_load = op.startswith(("buffer_load", "global_load", "scratch_load", "flat_load"))
        is_ds_read = op.startswith("ds_read")
        if touched & pending:
            findings.append((ln, s, sorted(touched & pending)[:4]))
        if op == "s_barrier" and any(x == "LDSDMA" for x in vm):
            findings.append((ln, s, [f"{sum(1 for x in vm if x == 'LDSDMA')} LDS-DMA piece(s) not waited for at the barrier"]))
        if is_vm_load:
            if " lds" in s:
                vm.append("LDSDMA")
            else:
                dst = args.split(",")[0]
                vm.append(regs(dst))
        elif op.startswith(("buffer_store", "global_store", "scratch_store")):
            vm.append(set())
        elif is_ds_read:
            lg.append(regs(args.split(",")[0]))
    return findings

def main():
    path = sys.argv[1]
    text = open(path).read().splitlines()
    # split into kernels
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN2y417conv_halo2_kernel.*:\s*(;.*)?$", l)]
    bad = 0
    for si, st in enumerate(starts):
        name = text[st].split(":")[0]
        end = next(i for i in range(st, len(text)) if "s_endpgm" in text[i])
        body = [(i + 1, text[i]) for i in range(st, end)]
        # the K loop region: from the first to the last MFMA
        mf = [k for k, (_, l) in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        # start a little before the first MFMA (the prologue's loads) -- from the first asm buffer_load
        first = next(k for k, (_, l) in enumerate(body) if "buffer_load_dwordx4" in l and " lds" not in l)
        # ... and on to the drain behind the loop: the loads still in flight at the loop's exit have dead destinations as far as the
        # compiler knows, and whatever it puts there before the `s_waitcnt vmcnt(0)` is overwritten when they land
        drain = next((k for k in range(mf[-1], len(body)) if re.search(r"s_waitcnt\s+vmcnt\(0\)", body[k][1])), mf[-1])
        region = body[first:mf[-1] + 1]
        # the chunk loop runs more than once: its body a second time behind the first pass (the state at the back edge -- loads still in
        # flight -- meets the loop's head again).  The loop = from the last label in front of the first MFMA to the backward branch to it.
        head = max((k for k in range(mf[0]) if re.match(r"^\.LBB\d+_\d+:", body[k][1])), default=None)
        if head is not None:
            label = body[head][1].split(":")[0]
            back = [k for k in range(mf[-1], len(body)) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", body[k][1])]
            if back:
                region = body[first:back[0] + 1] + body[head:back[0] + 1] + body[back[0] + 1:drain + 1]
        f = audit(region, name)
        nm = len(mf)
        print(f"{name[:70]}: {nm} MFMAs, {len(f)} finding(s)")
        for ln, s, r in f[:12]:
            print(f"    line {ln}: {s}   <- pending {r}")
        bad += len(f)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
