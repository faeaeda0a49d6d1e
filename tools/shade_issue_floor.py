"""k_shade's vector-ALU issue floor: the executed instructions of one launch by class (rocprofv3 SQ_INSTS_VALU_* counters,
profiles/<tag>_pmc_sq.json) priced with the issue cost of each class as measured on gfx950 by
tools/microbench/issue_rate.hip (profiles/r02_issue_rate.txt; cycles per wave64 instruction per SIMD with two or more
waves resident -- a single wave issues one instruction of any kind per ~4.4 cycles).
usage: python tools/shade_issue_floor.py profiles/r02_pmc_sq.json [kernel duration alone in us] > profiles/r02_k_shade_issue_floor.txt"""
import json, sys

COST = {  # class -> (cycles, which rows of r02_issue_rate.txt it stands on)
    "FMA_F32": (2.4, "fma_f32 2.37-2.41, fmac_f32 2.05, fmaak 2.40"),
    "MUL_F32": (2.1, "mul_f32 2.06-2.12"),
    "ADD_F32": (2.1, "add_f32 / sub_f32 2.09-2.13"),
    "CVT": (2.1, "iso_cvt_f32_ubyte1 2.1, iso_cvt_f32_i32 2.1 (4.0 back to back)"),
    "INT32": (3.0, "shifts / add 1.5-2.8, lshl_add 2.8, bfe 3.0, mul_lo_u32 3.8, and_or / add3 3.8-3.9"),
    "TRANS_F32": (13.2, "iso_rcp_f32 13.3, iso_rsq_f32 13.1"),
    "other": (2.0, "cmp 1.5-1.8, cndmask 2.0, max / min 1.7-2.0, mov 2.1; v_readfirstlane 9.4 (4 per wave)"),
}
SIMDS, GHZ = 256 * 4, 2.4

d = json.load(open(sys.argv[1]))
k = d["kernels"]["k_shade"]
total = k["SQ_INSTS_VALU"]
classes = {c[len("SQ_INSTS_VALU_"):]: v for c, v in k.items() if c.startswith("SQ_INSTS_VALU_")}
classes["other"] = total - sum(classes.values())
waves = k["SQ_WAVES"]
print(f"# k_shade, workload {d['workload']}, kernel sources {d['kernel_source_sha256'][:12]}: executed wave64 instructions per launch")
print(f"# {int(total)} VALU in {int(waves)} waves = {total / waves:.0f} per wave (one wave = 64 fragments); "
      f"SALU {int(k.get('SQ_INSTS_SALU', 0))}, SMEM {int(k.get('SQ_INSTS_SMEM', 0))}, VMEM_RD {int(k.get('SQ_INSTS_VMEM_RD', 0))}, "
      f"LDS {int(k.get('SQ_INSTS_LDS', 0))}, BRANCH {int(k.get('SQ_INSTS_BRANCH', 0))}")
print(f"{'class':10s} {'instructions':>13s} {'per wave':>9s} {'cycles each':>12s} {'M SIMD-cycles':>14s}   measured as")
cyc = 0.0
for c, n in sorted(classes.items(), key=lambda kv: -kv[1]):
    cost, rows = COST[c]
    cyc += n * cost
    print(f"{c:10s} {int(n):13d} {n / waves:9.1f} {cost:12.1f} {n * cost / 1e6:14.1f}   {rows}")
floor_us = cyc / SIMDS / (GHZ * 1e3)
ideal_us = total * 2.0 / SIMDS / (GHZ * 1e3)
print(f"issue floor at the measured costs: {cyc / 1e6:.1f} M SIMD-cycles / {SIMDS} SIMDs / {GHZ} GHz = {floor_us:.1f} us"
      f"   (every instruction at 2.0 cycles: {ideal_us:.1f} us)")
if len(sys.argv) > 2:
    t = float(sys.argv[2])
    print(f"measured alone: {t:.1f} us -> {floor_us / t:.2f} of the time is vector-ALU issue at the measured costs ({ideal_us / t:.2f} at 2.0 cycles)")
