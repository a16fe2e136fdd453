# gfx clock while the decode attention kernel runs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration,
# inside the bench step and in the isolated loop (tools/exp/decode_len_probe.py).  Separate --pmc pass, kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk/step -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_clk_step.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk/iso -- python3 tools/exp/decode_len_probe.py > gpurun_out/pmc_clk_iso.log 2>&1
python3 - <<'PY'
import csv, glob, statistics
for d in ("step", "iso"):
    cf = glob.glob(f"gpurun_out/pmc_clk/{d}/*/*counter_collection.csv")[0]
    kf = glob.glob(f"gpurun_out/pmc_clk/{d}/*/*kernel_trace.csv")[0]
    dur = {}
    for r in csv.DictReader(open(kf)):
        if "decode_mfma_pair" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    clk, us = [], []
    for r in csv.DictReader(open(cf)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            ns = dur[r["Dispatch_Id"]]
            clk.append(float(r["Counter_Value"]) / 8 / ns)  # cycles per ns = GHz
            us.append(ns / 1e3)
    print(f"{d}: {len(clk)} launches, duration median {statistics.median(us):.2f} us, gfx clock median {statistics.median(clk):.3f} GHz "
          f"(p10 {sorted(clk)[len(clk)//10]:.3f}, p90 {sorted(clk)[9*len(clk)//10]:.3f})")
PY
rm -rf gpurun_out/pmc_clk
