bash tools/gpu_r06.sh evidence r06
