#!/usr/bin/env python
"""Static check of csrc/conv_pws.hip's generated code: registers that an inline-asm `global_load_dwordx4` is still filling must
not be touched before the counted wait that retires them.

The residual of a tile is requested by inline-asm loads whose results the compiler believes to be valid at once; a copy it
chooses to insert (phi, tied operand, live-range split) in front of the `s_waitcnt vmcnt(N)` that retires the load reads stale
registers -- the race `profiles/r03/o_pws_race_screen_both_schedules.log` shows (first build: `o_pw_check_first_version.log`).
Nothing in the language forbids such a copy, so the build checks the assembly: for every kernel, load number k (unit k // 2 of
the epilogue) may be mentioned again only after k // 2 + 1 inline-asm waits (units requested all at once), or -- the GDN
instances, which request unit g + 1 before they wait for unit g -- after one more wait than the youngest unit still pending.

Limitation: the scan is LINEAR over the assembly text, not a control-flow analysis -- it follows the order of the file.  With
forward branches only, every execution path between a load and its wait is a sub-sequence of the text between them, so
checking all of that text is conservative; what a linear scan cannot see is rejected instead of trusted: a BACKWARD branch or
an indirect jump while loads are in flight, and a wait that a forward branch (taken while loads were in flight) could bypass.

    hipcc ... --save-temps=obj -c conv_pws.hip -o /tmp/x.o ; python tools/check_inflight_regs.py /tmp/conv_pws-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check_kernel(name, lines):
    """-> (problems, number of inline-asm loads)"""
    problems = []
    total = 0
    pending = []            # [registers, waits still needed]
    in_asm = False
    nload = 0
    unit_need = 1
    seen_labels, open_targets = set(), set()
    for ln, line in lines:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m_lab = re.match(r"^(\.LBB\d+_\d+):", t)
        if m_lab:
            seen_labels.add(m_lab.group(1))
            open_targets.discard(m_lab.group(1))
        if not t or t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0]
        if pending and re.match(r"^s_(setpc|call|swappc)", code):
            problems.append((ln, t, "an indirect jump while residual loads are in flight"))
        m_br = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", code)
        if pending and m_br:
            if m_br.group(1) in seen_labels:
                problems.append((ln, t, "a BACKWARD branch while residual loads are in flight (the scan follows file order)"))
            else:
                open_targets.add(m_br.group(1))        # forward: every path still runs through later text, but may skip a wait
        if in_asm and (code.startswith("global_load_dwordx4") or code.startswith("global_load_dwordx2")):
            dst = code.split()[1].rstrip(",")
            if not pending:
                nload = 0
                open_targets.clear()
            # the address operands of THIS load are read at issue: only the destination becomes in-flight
            used = regs_of(code.split(",", 1)[1])
            for regs, _ in pending:
                if regs & used:
                    problems.append((ln, t, "address uses an in-flight register"))
            # a unit = two consecutive loads; it is retired by one more wait than the youngest unit still pending (units
            # requested all at once need 1, 2, 3 ... waits; a unit requested one ahead of its predecessor's wait needs 2)
            if nload % 2 == 0:
                unit_need = max((p_[1] for p_ in pending), default=0) + 1
            pending.append([regs_of(dst), unit_need])
            nload += 1
            total += 1
            continue
        if in_asm and code.startswith("s_waitcnt") and "vmcnt" in code and pending:
            if open_targets:
                problems.append((ln, t, "a forward branch taken while loads were in flight may bypass this wait: %s" % sorted(open_targets)))
            for p in pending:
                p[1] -= 1
            pending = [p for p in pending if p[1] > 0]
            continue
        if pending:
            used = regs_of(code)
            for regs, _ in pending:
                if regs & used:
                    problems.append((ln, t, "touches registers v%s of a load that is still in flight" % sorted(regs & used)))
                    break
    return problems, total


def main():
    path = sys.argv[1]
    kernels, cur, name = {}, None, None
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            m = re.match(r"^(_Z\w*conv_pws_kernel\w*):", line)
            if m:
                name, cur = m.group(1), []
                kernels[name] = cur
                continue
            if cur is not None:
                cur.append((ln, line))
                if line.startswith(".Lfunc_end"):
                    cur = None
    bad = 0
    with_loads = 0
    for name, lines in kernels.items():
        probs, has = check_kernel(name, lines)
        with_loads += has > 0
        for ln, t, why in probs[:5]:
            print(f"{name}: line {ln}: {t}   <- {why}")
        bad += bool(probs)
    print(f"{len(kernels)} kernels, {with_loads} with inline-asm residual loads, {bad} with in-flight register hazards")
    sys.exit(1 if bad or not kernels else 0)


if __name__ == "__main__":
    main()
