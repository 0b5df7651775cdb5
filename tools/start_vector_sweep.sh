set -e
cd $GRAFT_REPO_ROOT
run() { # name newton_pts newton_deg press_pts press_deg corr_pts corr_deg
  python3 tools/long_run.py 500 newton.linear_start_points=$2 newton.linear_start_degree=$3 pressure.start_points=$4 pressure.start_degree=$5 correction.start_points=$6 correction.start_degree=$7 > gpurun_out/r4_sweep_$1.log 2>&1
  echo "$1: $(grep 'steps 20' gpurun_out/r4_sweep_$1.log)"
}
run n32_p32 3 0 3 0 3 0
run n43_p43 4 0 4 0 4 0
run n54_p43 5 0 4 0 4 0
run n64_p63 6 4 6 3 6 4
run n64_p53 6 4 5 3 5 3
run n53_p53 5 3 5 3 5 3
run n43_p63 4 0 6 3 4 0
