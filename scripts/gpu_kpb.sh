cd $GRAFT_REPO_ROOT
for k in 2048 4096 8192; do
  echo "KPB=$k"; GSR_SORT_KPB=$k SCRIPT="scripts/big_scene_check.py 20000000" bash scripts/gpu_prof_py.sh > /dev/null 2>&1
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("gpurun_out/profp/*/*kernel_stats.csv")[0])):
    n=r["Name"]
    if "k_scatter" in n or "hist" in n or "column_scan" in n: print("  %-34s calls %3s avg %9.1f us" % (n.split("(")[0].replace("void ","").replace("gsr::","")[:34], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
