cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_restoration.py -m gpu -x -q 2>&1 | tail -2
python tools/microbench.py lr 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin)
for k,v in j.items(): print(k, round(v['ms']*1000,1),'us')"
