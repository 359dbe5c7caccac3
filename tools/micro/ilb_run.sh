for v in base nolit nocopy nomem k1 k24; do for s in q; do echo "== $v $s 32768"; timeout 120 tools/micro/ilb_$v.bin tools/micro/sample_$s.bin 32768 2; done; done
echo "== base nq 32768"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_nq.bin 32768 2
echo "== base q 8192"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_q.bin 8192 2
echo "== base q 65536"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_q.bin 65536 2
