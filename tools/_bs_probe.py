import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, torch
import oracle
from elementary_amd import graphs, el
from elementary_amd.runtime import Runtime
from cases import every_stateful_roots
from helpers import lcg_noise
bs=int(sys.argv[1]) if len(sys.argv)>1 else 192
def run(spec, fn, n_out, n_in, batch):
    a=Runtime(48000.0,bs,device=0); a.set_option("specialize",spec); a.set_option("batch_blocks",batch)
    c=oracle.RefRuntime(48000.0,bs)
    assert a.render(*fn())["result"]==0 and c.render(*fn())["result"]==0
    nb=40
    x=np.stack([np.stack([lcg_noise(bs,7+k,0.5)]) for k in range(nb)]) if n_in else None
    out=torch.zeros((nb,n_out,bs),dtype=torch.float32,device="cuda")
    if n_in:
        xin=torch.from_numpy(x).cuda(); torch.cuda.synchronize()
        a.process_blocks(nb,n_out,out_ptr=out.data_ptr(),in_ptr=xin.data_ptr(),num_inputs=1)
    else:
        torch.cuda.synchronize(); a.process_blocks(nb,n_out,out_ptr=out.data_ptr())
    got=out.cpu().numpy()
    ref=np.stack([c.process(x[k] if n_in else None,n_out,bs) for k in range(nb)])
    err=np.abs(got-ref).max(axis=(1,2))
    print("spec",spec,"batch",batch,"max err",err.max(),"first bad block",int(np.argmax(err>1e-6)) if (err>1e-6).any() else None, a.stats()["spec_launches"], a.stats()["batch_launches"])
voice=lambda: [graphs.c2_voice(0)]
for spec in (0,2):
    for batch in (16,1):
        run(spec, lambda: graphs.c2_graph(voices=16), 2, 0, batch)
run(2, voice, 1, 0, 16)
run(2, lambda: [el.phasor(100.0)], 1, 0, 16)
run(2, lambda: [el.pole(0.99, el.phasor(100.0))], 1, 0, 16)
run(2, lambda: [el.blepsaw(100.0)], 1, 0, 16)
run(2, lambda: [el.svf({"mode":"lowpass"}, 800.0, 1.0, el.blepsaw(100.0))], 1, 0, 16)
