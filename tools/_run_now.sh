export K="collect or cigar or golden or entry or c1_ or dropin or config or edit_distance or scheduling or small or segment"
export VARIANTS="tree prev nu3" WL="c1"
export E2E="waves: waves8g:SVX_BAM_DEV_CHUNK_MB=8192 lanes8g:SVX_INFLATE_LANES=1,SVX_BAM_DEV_CHUNK_MB=8192 lanes8gq8:SVX_INFLATE_LANES=1,SVX_BAM_DEV_CHUNK_MB=8192,GPU_MAX_HW_QUEUES=8 lanes4q8:SVX_INFLATE_LANES=1,SVX_BAM_DEV_CHUNK_MB=8192,GPU_MAX_HW_QUEUES=8,SVX_BAM_DEV_SLOTS=4,SVX_BAM_DEV_SUB=16384 lanes6q8:SVX_INFLATE_LANES=1,SVX_BAM_DEV_CHUNK_MB=8192,GPU_MAX_HW_QUEUES=8,SVX_BAM_DEV_SLOTS=6,SVX_BAM_DEV_SUB=11000 wavesq8:GPU_MAX_HW_QUEUES=8"
bash tools/gpu_r06.sh quick r06z
