#!/usr/bin/env python3
"""One table row per streaming kernel from the summary.txt files tools/pmc_cmd.sh writes.

usage: python tools/summarize_pmc.py <symbols per launch> <summary.txt> [<summary.txt> ...]
Columns: average duration, HBM bytes from the counters (FETCH_SIZE counts gfx950's 128-byte
requests as 64 bytes, so the read side is doubled; WRITE_SIZE is exact; both in KiB), vector
instructions per symbol and lane, VALU busy = vector instructions x 4 cycles over the SIMD-cycles
of the launch (1024 SIMDs, cycles from GRBM_GUI_ACTIVE over the 8 XCDs), share of wave-cycles
spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), LDS bank-conflict cycles per LDS-active cycle.
"""
import sys


def rows(path):
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        name, *kv = line.split()
        yield name, {k: float(v) for k, v in (x.split("=") for x in kv)}


def main():
    nsym = float(sys.argv[1])
    print("| probe | kernel | avg us | FETCHx2 GB | WRITE GB | VALU/sym | SALU/sym | LDS/sym | "
          "VALU busy | waiting | LDS conflict |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for path in sys.argv[2:]:
        probe = path.split("/")[-2]
        for name, c in rows(path):
            if not any(k in name for k in ("encode", "decode", "hist")) or "_avg_ns" not in c:
                continue
            disp = c.get("_dispatches", 1.0)
            steps = nsym / 64.0 * disp
            cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            busy = c.get("SQ_INSTS_VALU", 0.0) * 4.0 / (1024.0 * cyc) if cyc else float("nan")
            conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0))
            wait = c.get("SQ_WAIT_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0))
            print(f"| {probe} | `{name}` | {c['_avg_ns'] / 1e3:.0f} | "
                  f"{2 * c.get('FETCH_SIZE', 0) * 1024 / 1e9 / disp:.3f} | "
                  f"{c.get('WRITE_SIZE', 0) * 1024 / 1e9 / disp:.3f} | "
                  f"{c.get('SQ_INSTS_VALU', 0) / steps:.1f} | {c.get('SQ_INSTS_SALU', 0) / steps:.1f} | "
                  f"{c.get('SQ_INSTS_LDS', 0) / steps:.2f} | {busy:.2f} | {wait:.2f} | {conf:.2f} |")


if __name__ == "__main__":
    main()
