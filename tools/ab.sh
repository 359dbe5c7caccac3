#!/bin/bash
# interleaved A/B of libsvx variants on the workloads given: tools/ab.sh "c1 c2" prev tree ...   (a variant may carry environment settings: tree@SVX_MAILBOX=0,SVX_X=1)
wl=$1; shift
for rep in 1 2; do for w in $wl; do for v in "$@"; do
  name=${v%%@*}; envs=""; [ "$name" != "$v" ] && envs=$(echo ${v#*@} | tr ',' ' ')
  lib=svim_amd/variants/libsvx_$name.so; [ "$name" = "tree" ] && lib=svim_amd/libsvx.so
  env $envs SVX_LIB=$PWD/$lib python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-end-to-end --workload $w 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels']; print(sys.argv[1], sys.argv[2], 'ms %.2f scan %.3f collect %.2f part %.2f edit %.2f link %.2f wc %.4g retry %.3g' % (j['ms_per_step'], k['k_cigar_scan_ms'], k['collect_total_ms'], k['cluster_partition_sample_ms'], k['cluster_edit_distance_ms'], k['cluster_linkage_ms'], k['edit_wordcols_issued'] or 0, k['edit_wordcols_retry_rounds'] or 0))" $w $v
done; done; done
