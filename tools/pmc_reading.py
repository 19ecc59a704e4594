#!/usr/bin/env python3
"""Condensed per-kernel view of tools/collect_pmc.sh's output (pmc.json + resources.json): what each batched kernel does with
its wave slots. Usage: tools/pmc_reading.py <dir or pmc.json> [resources.json]

Columns (per launch, means over the launches of the run; counters are collected with the kernels SERIALISED by the profiler):
  waves     SQ_WAVES
  qc/wave   SQ_WAVE_CYCLES / SQ_WAVES  - wave lifetime in quad-cycles (SQ_WAVE_CYCLES counts 4-cycle units on gfx950)
  Mqc       SQ_WAVE_CYCLES / 1e6       - wave-slot occupancy of the launch
  valu/w, vmem/w   instructions per wave
  wait%     SQ_WAIT_ANY / SQ_WAVE_CYCLES       - share of its lifetime a wave waits for anything (memory, LDS, barrier, sleep)
  iss%      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES  - waits for an issue slot
  act%      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  hit%      TCC_HIT / (TCC_HIT + TCC_MISS)
  fetchMB / writeMB   2 x FETCH_SIZE, WRITE_SIZE (KiB counters, gfx950 correction of MI355X_MICROARCH.md) per launch
  us(ser)   GRBM_GUI_ACTIVE / 8 XCDs / 2.1 GHz - duration of the launch when it runs ALONE
  slots%    Mqc * 4 / (us(ser) * 2100 cycles * 1024 SIMDs * 8 slots) - how full the chip's wave slots are while it runs alone
"""
import json
import os
import sys


def main():
    a = sys.argv[1]
    pj = os.path.join(a, "pmc.json") if os.path.isdir(a) else a
    rj = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(pj), os.path.basename(pj).replace("pmc.json", "resources.json"))
    d = json.load(open(pj))
    r = json.load(open(rj)) if os.path.exists(rj) else {}
    ks = [k for k in d if k.startswith("k_")]

    def g(k, c):
        return d[k].get(c, {}).get("mean", float("nan"))

    print("%-26s %4s | %7s %8s %6s %6s %6s | %5s %5s %5s | %5s %7s %7s | %7s %6s" % (
        "kernel", "vgpr", "waves", "qc/wave", "Mqc", "valu/w", "vmem/w", "wait%", "iss%", "act%", "hit%", "fetchMB", "writeMB", "us(ser)", "slots%"))
    for k in sorted(ks, key=lambda k: -g(k, "SQ_WAVE_CYCLES")):
        w = g(k, "SQ_WAVES")
        wc = g(k, "SQ_WAVE_CYCLES")
        if not w or w != w:
            continue
        us = g(k, "GRBM_GUI_ACTIVE") / 8 / 2.1e3
        print("%-26s %4s | %7.0f %8.0f %6.1f %6.0f %6.0f | %5.1f %5.1f %5.1f | %5.1f %7.2f %7.2f | %7.1f %6.1f" % (
            k[:26], r.get(k, {}).get("VGPR_Count", "?"), w, wc / w, wc / 1e6, g(k, "SQ_INSTS_VALU") / w, g(k, "SQ_INSTS_VMEM") / w,
            100 * g(k, "SQ_WAIT_ANY") / wc, 100 * g(k, "SQ_WAIT_INST_ANY") / wc, 100 * g(k, "SQ_ACTIVE_INST_ANY") / wc,
            100 * g(k, "TCC_HIT_sum") / (g(k, "TCC_HIT_sum") + g(k, "TCC_MISS_sum")), 2 * g(k, "FETCH_SIZE") / 1024, g(k, "WRITE_SIZE") / 1024,
            us, 100 * wc * 4 / (us * 2100 * 1024 * 8)))


if __name__ == "__main__":
    main()
