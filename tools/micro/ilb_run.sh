for v in base nocopy nomem k16; do echo "== $v q 65536"; timeout 120 tools/micro/ilb_$v.bin tools/micro/sample_q.bin 65536 2; done
echo "== base q 32768"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_q.bin 32768 2
echo "== base q 16384"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_q.bin 16384 2
echo "== base nq 65536"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_nq.bin 65536 2
