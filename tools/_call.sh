cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_fix1; mkdir -p $O
nproc > $O/nproc.txt
B=oracle/_ref/fixtures/SvtAv1HipFixtures
ldd $B | grep -i "not found" > $O/ldd_missing.txt
N=8
for i in $(seq 0 $((N-1))); do
  ( GTEST_TOTAL_SHARDS=$N GTEST_SHARD_INDEX=$i timeout 1500 $B --gtest_filter='HIP*' --gtest_output=json:$O/shard$i.json > $O/shard$i.txt 2>&1; echo "rc=$?" >> $O/shard$i.txt ) &
done
wait
tail -3 $O/shard*.txt
