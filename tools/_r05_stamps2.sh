#!/bin/bash
cd $GRAFT_REPO_ROOT
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
export DCF_HIP_LIB=$PWD/$P/libdcf_hip_vstamp.so
DCF_STAMP_RAW=1 python3 tools/chain_stamps.py 2x44x50x256 3 2>&1 | grep -v amdgpu.ids | head -40
unset DCF_HIP_LIB
python3 - <<'PY'
import importlib, torch, time
PKG="deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops=importlib.import_module(PKG+".ops"); H=importlib.import_module(PKG+"._hip")
def run(shape,n,chain):
    B,Hh,W,C=shape
    x=(torch.rand((B,Hh,W,C),device="cuda")-0.5).bfloat16(); ext=(torch.rand((B,Hh,W,C),device="cuda")-0.5).bfloat16()
    wts=[((torch.rand((C,3,3,C),device="cuda")-0.5)*0.05).bfloat16() for _ in range(n)]
    ws=ops.conv3x3_chain_workspace(1,B,Hh,W,C,n,"cuda")
    layers=[(wts[l],None,(ext if l==0 else l-2) if l%2==0 else None,None,True) for l in range(n)]
    def sep():
        cur=x; outs=[]
        for l,(w,sh,r,m,relu) in enumerate(layers):
            rr = outs[r] if type(r) is int else r
            cur=ops.conv2d_fwd(1,cur,w,None,rr,3,3,1,1,relu,C); outs.append(cur)
    f=(lambda: ops.conv3x3_chain(1,x,layers,0,ws)) if chain else sep
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/20/n
for shape,n in (((2,44,50,256),11),((2,88,100,192),11),((2,176,200,128),7),((1,44,50,256),11),((1,88,100,192),11),((1,176,200,128),7)):
    print(shape,n,"us/layer separate %.2f chain %.2f"%(run(shape,n,False),run(shape,n,True)))
PY
