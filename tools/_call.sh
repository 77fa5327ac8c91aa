cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call6; mkdir -p $O
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== init timing"; SVT_HIP_INIT_TIMING=1 timeout 300 $E --case fps_1080p_p8_all_tplrecon > $O/init.log 2>&1; grep -a "encoder fps" $O/init.log
python - <<'PY'
import glob
# the HIP run's stderr is not kept by enc_identity: re-run the encoder directly for the timing lines
PY
python tools/enc_identity.py --help > /dev/null
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, 'tools')
import enc_identity as ei
w,h,n,bd=1920,1080,10,8
os.makedirs('/tmp/idt', exist_ok=True)
clip='/tmp/idt/init.yuv'; ei.make_clip(clip,w,h,n,bd)
lib=os.path.join(ei.ROOT,'svt-av1-psy_amd','libsvtav1_hip.so')
env=dict(os.environ); env.update(ei.seam_env('fps_1080p_p8_all_tplrecon', lib, '/tmp/idt', 'init')); env['SVT_HIP_INIT_TIMING']='1'
for tag,e in (('avx2 alone', dict(os.environ)), ('avx2 + stages', env)):
    t0=time.time()
    r=subprocess.run([ei.ENC_AVX2,'-i',clip,'-w',str(w),'-h',str(h),'--fps','30','-n',str(n),'--input-depth','8','--preset','8','-b','/tmp/idt/init.ivf'],capture_output=True,text=True,env=e)
    print(tag, 'wall %.3f s for %d frames' % (time.time()-t0, n))
    for ln in (r.stdout+r.stderr).splitlines():
        if 'INIT_TIMING' in ln or 'Average Speed' in ln: print('   ', ln.strip())
PY
echo "== instances 4, 300-frame clip"; timeout 600 $E --case fps_1080p_p8_all_tplrecon_300 --instances 4
echo "== instances 8, 300-frame clip"; timeout 600 $E --case fps_1080p_p8_all_tplrecon_300 --instances 8
echo "== preset 10, 300 frames"; timeout 600 $E --case fps_1080p_p10_all_tplrecon_300 2>&1 | grep -a "identical=\|encoder fps" | cut -c1-60,1400-1700
