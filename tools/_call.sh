cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call2; mkdir -p $O
# 1. the reference fixtures through the pytest (default mode), timed
( time timeout 1200 python -m pytest tests/test_ref_fixtures.py -q -m gpu -x ) > $O/pytest_fixtures.txt 2>&1; tail -5 $O/pytest_fixtures.txt
# 2. the suite that failed in call 1 (highbd statistics at bit depth 8), in full
B=oracle/_ref/fixtures/SvtAv1HipFixtures
for i in 0 1 2 3 4 5 6 7; do ( GTEST_TOTAL_SHARDS=8 GTEST_SHARD_INDEX=$i timeout 600 $B --gtest_filter='HIP/av1_compute_stats_test_hbd*' > $O/stats_hbd_$i.txt 2>&1 ) & done; wait
grep -h "PASSED\|FAILED TEST" $O/stats_hbd_*.txt | sort | uniq -c
# 3. strict 10-bit identity: every 10-bit GPU case three times, first attempt decides
T10=p8_10bit_lp1,p4_10bit_lp4,p8_10bit_lossless,seam_p8_10bit,lrseam_p2_10bit,lrseam_p3_10bit_crf50,cdefseam_p4_10bit,dlfseam_p2_10bit,dlfseam_sb_p10_10bit,tfseam_p4_10bit,tfdriver_p8_10bit,tfdriver_p4_10bit,lowdelay_720p_p10_10bit,tfsubpel_p2_10bit,tplseam_p8_10bit,vstrips2_cdef_lr_p8_10bit
for rep in 1 2 3; do
  timeout 900 python tools/enc_identity.py --case $T10 --out $O/id10_$rep > $O/identity10_$rep.txt 2>&1; tail -1 $O/identity10_$rep.txt
done
timeout 600 python tools/enc_identity.py --case everyseam_4k10_p8_lp1 --out $O/id10_4k > $O/identity10_4k.txt 2>&1; tail -1 $O/identity10_4k.txt
rm -rf $O/id10_*/*.yuv $O/id10_*/*.ivf
# 4. the bench line with the new legs
( time timeout 1500 python bench.py ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -c 1500 $O/bench_stdout.txt; cp gpurun_out/bench_detail.json $O/ 2>/dev/null
tail -5 $O/bench_stderr.txt | cut -c1-600
