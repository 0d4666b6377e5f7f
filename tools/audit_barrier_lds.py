#!/usr/bin/env python3
"""Audit of the raw-`s_barrier` slot hand-overs in the LDS-DMA ring kernels (Weak #2 / ADVICE round 4).

The ring kernels hand an LDS slot back to the DMA behind a raw `s_barrier` (no `s_waitcnt` of the compiler's own in front of it,
unlike __syncthreads()).  That is only safe if every LDS READ a wave issued from the slot has RETURNED when the wave arrives at
the barrier -- otherwise another wave can pass the barrier and aim an LDS-DMA piece at the slot while this wave's ds_read is
still queued (conv_lc.hip had exactly that; fixed with an explicit `s_waitcnt lgkmcnt(0)`).

This tool disassembles the objects and walks every kernel linearly: an LDS read (`ds_read*` / `ds_load*`) marks the LGKM counter
as possibly non-zero; an `s_waitcnt` whose lgkmcnt field is 0 clears the mark; an `s_barrier` reached with the mark set is
REPORTED.  (Linear order = program order inside a loop body; a wait on the loop's back edge is seen when the body is re-entered
textually, so the first barrier of a loop body is checked against the instructions in front of the loop -- conservative.)

    python tools/audit_barrier_lds.py [objects...]        (default: the four ring-kernel objects + conv_lc)
Exit status 1 when a barrier with LDS reads possibly in flight is found."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CLANG_OFFLOAD_BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def device_code(obj):
    """The gfx950 code object inside a host object's .hip_fatbin section -> temp file path."""
    fat, out = obj + ".fatbin.tmp", obj + ".gfx950.co"
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    try:
        subprocess.check_call([CLANG_OFFLOAD_BUNDLER, "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], stderr=subprocess.DEVNULL)
    finally:
        os.remove(fat)
    return out


def audit(obj):
    co = device_code(obj)
    try:
        txt = subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", co], text=True)
    finally:
        os.remove(co)
    kernel, pending, last_read = None, False, None
    findings, barriers, kernels = [], 0, 0
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel, pending = m.group(1), False
            kernels += 1
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if op.startswith(("ds_read", "ds_load")):
            pending, last_read = True, ins
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m and int(m.group(1)) == 0:
                pending = False
            elif not re.search(r"vmcnt|expcnt|lgkmcnt", ins):       # raw immediate form: decode the lgkm field (bits 11:8)
                m2 = re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)", ins)
                if m2 and ((int(m2.group(1), 0) >> 8) & 0xF) == 0:
                    pending = False
        elif op == "s_barrier":
            barriers += 1
            if pending:
                findings.append((kernel, last_read))
    return kernels, barriers, findings


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    objs = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("conv_rs.o", "conv_chain.o", "conv_wgs.o", "conv_wg1.o", "conv_sp.o", "conv_lc.o", "conv_wgv.o", "conv.o")]
    bad = 0
    for o in objs:
        kernels, barriers, findings = audit(o)
        per = {}
        for k, r in findings:
            per.setdefault(k, []).append(r)
        print("%s: %d kernels, %d s_barrier, %d with LDS reads possibly in flight" % (os.path.basename(o), kernels, barriers, len(findings)))
        for name, dm in zip(per, demangle(list(per))):
            print("   %3d  %s" % (len(per[name]), dm[:160]))
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
