#!/usr/bin/env python
"""Run a command under the guard-page allocator and, if the GPU faults, name the buffer whose guard was hit.

    python tests/guard_alloc/run.py [--mode back|front] [--timeout S] [--log FILE] -- python -m pytest tests/... -x -q

Exit code = the command's.  On "Memory access fault by GPU ... on address 0x..." (the HSA runtime aborts the process) the
allocation log written by libguard_alloc.so is searched for the live mappings next to the address: an address in the
guard right AFTER a mapping is an over-read / over-write of that buffer by (address - end) bytes (page granular: the
runtime reports the page), right BEFORE it an under-run; an address inside a freed mapping is a use after free."""
import argparse
import os
import re
import subprocess
import sys

FAULT = re.compile(r"Memory access fault by GPU.*?on address (0x[0-9a-fA-F]+)", re.S)


def lookup(log_path, addr, window=4):
    live, freed = {}, {}
    with open(log_path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "A":
                ptr, size, lo, hi = int(p[1], 16), int(p[2]), int(p[3], 16), int(p[4], 16)
                live[ptr] = (size, lo, hi, len(p) > 5)
                freed.pop(ptr, None)
            elif p[0] == "F":
                ptr = int(p[1], 16)
                if ptr in live:
                    freed[ptr] = live.pop(ptr)
    out = []
    rows = sorted((lo, hi, ptr, size, cap, "live") for ptr, (size, lo, hi, cap) in live.items())
    rows += sorted((lo, hi, ptr, size, cap, "FREED") for ptr, (size, lo, hi, cap) in freed.items())
    near = sorted(rows, key=lambda r: min(abs(addr - r[0]), abs(addr - r[1])))[:window]
    for lo, hi, ptr, size, cap, state in near:
        if lo <= addr < hi:
            where = f"INSIDE the mapping ({state})"
        elif addr >= hi:
            where = f"{addr - hi} bytes past the END of the mapping (buffer ends at the mapping's end: over-run)"
        else:
            where = f"{lo - addr} bytes before the START of the mapping (under-run)"
        out.append(f"  buffer {ptr:#x} size {size} B  mapping [{lo:#x}, {hi:#x}) {state}{' capture' if cap else ''}: {where}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="back", choices=("back", "front", "plain"))
    ap.add_argument("--timeout", type=float, default=900.0)
    ap.add_argument("--log", default="/tmp/dgcn_guard.log")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--align", type=int, default=16)
    ap.add_argument("--free", default="keepva", choices=("unmap", "keepva", "never"))
    ap.add_argument("--no-blocking", action="store_true", help="do not serialise kernel launches (default: every launch "
                    "is waited for, so that the Python traceback printed on the abort names the faulting call)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    if not cmd:
        ap.error("no command")
    env = dict(os.environ, DGCN_GUARD_ALLOC="1", DGCN_GUARD_LOG=args.log, DGCN_GUARD_MODE=args.mode,
               DGCN_GUARD_POISON="1" if args.poison else "0", PYTHONFAULTHANDLER="1",
               DGCN_GUARD_ALIGN=str(args.align), DGCN_GUARD_FREE=args.free)
    if not args.no_blocking:
        env.update(HIP_LAUNCH_BLOCKING="1", AMD_SERIALIZE_KERNEL="3")
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.timeout)
        text, rc = r.stdout, r.returncode
    except subprocess.TimeoutExpired as exc:
        text = (exc.stdout or b"").decode(errors="replace") if isinstance(exc.stdout, bytes) else (exc.stdout or "")
        text += f"\n[guard run] TIMEOUT after {args.timeout:.0f} s\n"
        rc = 124
    sys.stdout.write(text[-20000:])
    m = FAULT.search(text)
    if m:
        addr = int(m.group(1), 16)
        print(f"\n[guard run] GPU memory fault at {addr:#x}; nearest guarded buffers:")
        for line in lookup(args.log, addr):
            print(line)
    print(f"[guard run] exit code {rc}")
    return rc


if __name__ == "__main__":
    sys.exit(main())
