for v in base d2 d6 nocopy; do echo "== $v q 65536"; timeout 120 tools/micro/ilb_$v.bin tools/micro/sample_q.bin 65536 2; done
echo "== base q 32768"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_q.bin 32768 2
echo "== base nq 65536"; timeout 120 tools/micro/ilb_base.bin tools/micro/sample_nq.bin 65536 2
