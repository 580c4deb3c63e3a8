import sys,os,tempfile; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from _pkg import load_package; load_package()
import numpy as np
from gpu_ai_inference_server_amd import binding as B
from gpu_ai_inference_server_amd.modelgen import models
from oracle import onnx_oracle as O, fp8 as F
def rel(a,b): return float(np.abs(np.asarray(a,np.float64)-b).max()/np.abs(b).max())
tmp=tempfile.mkdtemp()
for layers in ((1,0,0,0),(1,1,0,0),(2,1,2,1)):
    lay=tuple(l for l in layers if l)
    mb = models.resnet(3, layers=lay, width=16, image=64, classes=20, seed=51)
    path = models.write_repo(tmp, "r%d"%len(lay), mb)
    x = models.synthetic_input((3, 3, 64, 64), stream="resnet_f8")
    ref = O.run(O.load_model(mb), {"data": x}, dtype=np.float64)["logits"]
    os.environ["IE_PRECISION"]="fp8"
    m=B.CreateModel(path,"r"); info=B.RuntimeInfo(m)
    outs=[B.OutputConfig("logits",[3,20])]
    y=m.Infer([B.TensorData("data",B.DataTypeFloat32,B.Shape([3,3,64,64]),x)],outs)[0].Data.reshape(3,20).copy()
    m.Destroy()
    plan=B.DescribeModel(path,3)["plan"]; blob=B.PlanWeights(path,3)
    del os.environ["IE_PRECISION"]
    sc=info["f8_act_scales"]
    emu=F.run_plan(plan,blob,{"data":x},act_scales=sc,fp8=True)["logits"]
    print(lay,"engine vs emu",rel(y,emu),"engine vs ref",rel(y,ref),"emu vs ref",rel(emu,ref), flush=True)
    # variant: no weight quantisation
    import copy
    p2=copy.deepcopy(plan)
    for s in p2["steps"]:
        if s.get("algo")=="igemm_f8": s["algo"]="x"
    emu2=F.run_plan(p2,blob,{"data":x},act_scales=sc,fp8=True)["logits"]
    print("   no-weight-quant emu vs engine",rel(y,emu2))
    # variant: no activation quantisation
    p3=copy.deepcopy(plan)
    for s in p3["steps"]: s["out"]["f8"]=False
    emu3=F.run_plan(p3,blob,{"data":x},act_scales=sc,fp8=True)["logits"]
    print("   no-act-quant emu vs engine",rel(y,emu3))
    print("   scales",[round(v,5) for v in sc])
