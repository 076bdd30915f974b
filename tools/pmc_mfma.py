"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (own pass, kernel-trace only, program directly after `--`):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
              --kernel-trace -d gpurun_out/x -o mf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile
    python tools/pmc_mfma.py gpurun_out/x/mf_results.db [substring ...] > profiles/r04_pmc_mfma.txt
Per kernel (name cut at the argument list, grouped with its grid): launches, average duration from the dispatch timestamps, and
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock)      -- MI355X_MICROARCH.md: the counter counts cycles
              (32 per v_mfma_f32_32x32x16_bf16, ~16 per 16x16x32), summed over the chip's 256 CUs x 4 SIMDs
  clock     = GRBM_GUI_ACTIVE / duration when that counter was collected (the guide's "effective clock" recipe), else 2.4 GHz (max clock:
              the utilisation is then a LOWER bound)
  implied TFLOP/s = busy cycles x 1024 FLOP per busy cycle (a bf16 MFMA retires 16384 / 32768 FLOP per 16 / 32 cycles) / duration
Wave-level split (quad-cycles, disjoint): parked (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY) as
fractions of SQ_WAVE_CYCLES."""
import collections, re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2:] or ["gemm_tiled", "gemm_skinny", "attn", "gemm_f32"]
rows = db.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, value, start, end from counters_collection").fetchall()
disp = {}
for did, n, gx, gy, wx, c, v, s, e in rows:
    n = re.sub(r"\(.*", "", n)[:64]
    if not any(f in n for f in flt):
        continue
    d = disp.setdefault(did, dict(name=n, grid=f"{gx // max(wx, 1)}x{gy}", dur=e - s, c={}))
    d["c"][c] = d["c"].get(c, 0.0) + v
agg = collections.defaultdict(list)
for d in disp.values():
    agg[(d["name"], d["grid"])].append(d)
print(f"# {sys.argv[1]}: {len(disp)} dispatches matching {flt}")
print(f"{'kernel':64s} {'grid':>9s} {'n':>5s} {'avg_us':>8s} {'clk_GHz':>8s} {'mfma_util':>9s} {'impl_TF/s':>9s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s}")
for (n, g), ds in sorted(agg.items(), key=lambda kv: -sum(d["dur"] for d in kv[1])):
    dur = sum(d["dur"] for d in ds)                      # ns
    get = lambda k: sum(d["c"].get(k, 0.0) for d in ds)
    busy, gui, wc = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("GRBM_GUI_ACTIVE"), get("SQ_WAVE_CYCLES")
    clk = gui / dur if gui > 0 else 2.4                  # cycles per ns = GHz
    if clk > 2.4:                                        # GRBM_GUI_ACTIVE also covers the dispatch's ramp outside the kernel's own stamps: the
        clk = 2.4                                        # ratio then exceeds the 2.4 GHz maximum clock -- capped (utilisation = a lower bound)
    util = busy / (1024.0 * dur * clk) if dur > 0 else 0.0
    tf = busy * 1024.0 / dur / 1e3 if dur > 0 else 0.0   # FLOP / ns = GFLOP/s -> TFLOP/s
    frac = lambda k: (get(k) / wc) if wc > 0 else float("nan")
    print(f"{n:64s} {g:>9s} {len(ds):5d} {dur / len(ds) / 1e3:8.2f} {clk:8.2f} {util:9.3f} {tf:9.1f} {frac('SQ_WAIT_ANY'):7.2f} {frac('SQ_WAIT_INST_ANY'):8.2f} {frac('SQ_ACTIVE_INST_ANY'):8.2f}")
