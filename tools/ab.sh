#!/bin/bash
# interleaved A/B of libsvx variants on the workloads given: tools/ab.sh "c1 c2" prev tree ...
wl=$1; shift
for rep in 1 2; do for w in $wl; do for v in "$@"; do
  lib=svim_amd/variants/libsvx_$v.so; [ "$v" = "tree" ] && lib=svim_amd/libsvx.so
  SVX_LIB=$PWD/$lib python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-end-to-end --workload $w 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels']; print(sys.argv[1], sys.argv[2], 'ms %.2f collect %.2f part %.2f edit %.2f link %.2f wc %.4g retry %.3g' % (j['ms_per_step'], k['collect_total_ms'], k['cluster_partition_sample_ms'], k['cluster_edit_distance_ms'], k['cluster_linkage_ms'], k['edit_wordcols_issued'] or 0, k['edit_wordcols_retry_rounds'] or 0))" $w $v
done; done; done
