"""AddressSanitizer + UBSan pass over the CPU oracle (sanitizers are CPU-only on this pool).
usage: python tools/oracle_asan.py   (builds /tmp/liboracle_asan.so, re-runs itself under LD_PRELOAD=libasan)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ORA_ASAN_CHILD") != "1":
    so = "/tmp/liboracle_asan.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared", "-o", so, os.path.join(ROOT, "oracle", "ora_ops.c"), "-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2", ORA_ASAN_CHILD="1")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'instancesegmentation-jittor_amd')]
from oracle import ora
ora._SO = '/tmp/liboracle_asan.so'
ora._lib = ctypes.CDLL(ora._SO); ora._lib.ora_expf.restype = ctypes.c_float; ora._lib.ora_expf.argtypes=[ctypes.c_float]
import numpy as np
rng=np.random.default_rng(0)
x=rng.standard_normal((2,19,23,32)).astype(np.float32); w=(rng.standard_normal((40,3,3,32))*.1).astype(np.float32)
ora.conv2d(x,w,2,1,None,None,None,1); ora.conv2d(x,w[:,:1,:1],1,0)
ora.maxpool(x,3,2,1); ora.resize_bilinear(x,35,41,relu=1); ora.softmax(x)
b=np.abs(rng.standard_normal((300,4))).astype(np.float32)*50; b[:,2:]+=b[:,:2]; s=rng.uniform(0,1,300).astype(np.float32)
ora.nms(b,s,0.5); ora.topk(s,100); ora.level_map(b)
f=rng.standard_normal((1,25,42,8)).astype(np.float32); r=np.concatenate([np.zeros((30,1),np.float32),b[:30]],1); ora.roi_align(f,r,0.125,7,7,2)
lg=rng.standard_normal((300,81)).astype(np.float32); rg=rng.standard_normal((300,324)).astype(np.float32)*.1
ora.box_postprocess(lg,rg,b,1333,800)
ora.rpn_level(rng.standard_normal(40*56*3).astype(np.float32), rng.standard_normal((40*56*3,4)).astype(np.float32)*.1, np.abs(rng.standard_normal((40*56*3,4))).astype(np.float32)*30, 1000,1000,0.7,0,448,320)
P=500; conf=rng.standard_normal((P,81)).astype(np.float32); conf[:20,5]+=8
pri=np.concatenate([rng.uniform(.1,.9,(P,2)),rng.uniform(.05,.4,(P,2))],1).astype(np.float32); loc=rng.standard_normal((P,4)).astype(np.float32)*.3; msk=np.tanh(rng.standard_normal((P,32))).astype(np.float32)
d=ora.yolact_detect(ora.softmax(conf), ora.yolact_decode(loc,pri), msk)
ora.yolact_masks(np.abs(rng.standard_normal((24,24,32))).astype(np.float32), d['mask'], d['box'], 50, 61)
ora.paste_masks(rng.uniform(0,1,(3,28,28)).astype(np.float32), np.array([[-5,-4,30,40],[10,10,10.2,10.1],[100,50,170,130]],np.float32), 120, 160)
ora.deconv2x2(x, rng.standard_normal((32,6,2,2)).astype(np.float32), np.zeros(6,np.float32), 1)
ora.mask_logits_select(rng.standard_normal((3,784,64)).astype(np.float32), rng.standard_normal((81,64)).astype(np.float32), np.zeros(81,np.float32), np.array([1,5,80],np.int32))
print("asan/ubsan run complete")
