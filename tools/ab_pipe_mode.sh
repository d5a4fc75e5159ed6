for mode in 0 1 2 3; do
  LSLAM_EXP_PIPE_MODE=$mode python bench.py --steps 200 --no-cpu --no-secondary --sustained-s 0 > gpurun_out/pm_$mode.json 2> gpurun_out/pm_$mode.err || echo "mode $mode failed"
done
