bash tools/gpu_r06.sh suite r06G
bash tools/gpu_r06.sh bench r06G | cut -c1-600
