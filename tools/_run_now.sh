RANKS="8" bash tools/gpu_r06.sh strong r06J 2>&1 | tail -12
