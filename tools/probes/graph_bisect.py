"""Which part of a --norm batch iteration breaks hipGraph capture (segfault in hipStreamEndCapture)?  Each stage is captured in
a child process of its own: python tools/probes/graph_bisect.py [stage]

Found (round 6): the forward alone captures, anything with the BatchNorm BACKWARD in it crashes - unless the previous eager
iteration's autograd graph is dropped before the capture (BISECT_CLEAR=1, the default here; BISECT_CLEAR=0 reproduces the
crash).  The norm scales' / shifts' gradients are the only ones that still reach autograd's AccumulateGrad nodes (the conv shims
add theirs in place); a leaf's AccumulateGrad node is bound to the stream it was created on and is re-used while the old graph
is alive (MMHandModel keeps fake_nhwc and the generator's loss terms), so the captured backward ran it on the eager iterations'
stream: a fork out of the capturing stream.  MMHandModel._capture_step now clears those references first."""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STAGES = ["G_bwd", "all"]


def child(stage, norm):
    import torch
    from oracle import mmhand_ref as O
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    opt = default_train_opt(batchSize=2, ngf=8, ndf=8, n_layers_D=2, G_n_blocks=2, norm=norm, pool_size=3, name="gb",
                            checkpoints_dir="/tmp/mmh_gb", local_rank=0, graph_step=True)
    os.environ["MMH_GRAPH_CAPTURE"] = "0"
    random.seed(1)
    m = MMHandModel(opt)
    m.set_input(O.synthetic_batch(2, 32, 32, seed=1))
    for _ in range(3):
        m.optimize_parameters()
    torch.cuda.synchronize()
    for pool in (m.fake_PP_pool, m.fake_PB_pool):
        pool.begin_iteration(2, m.device)

    def body():
        ops.ACCUM_PARAM_GRADS = True
        if stage == "G_fwd_nograd":
            with torch.no_grad():
                m.forward()
            return
        m.forward()
        if stage == "G_fwd":
            return
        m.optimizer_G.zero_grad()
        m.backward_G()
        if stage == "G_bwd":
            return
        m._guarded_step(m.optimizer_G, 0, 0)
        if stage == "G_step":
            return
        if stage in ("D_PP", "all"):
            m.optimizer_D_PP.zero_grad(); m.backward_D_PP(); m._guarded_step(m.optimizer_D_PP, 1, 2)
        if stage in ("D_PB", "all"):
            m.optimizer_D_PB.zero_grad(); m.backward_D_PB(); m._guarded_step(m.optimizer_D_PB, 2, 1)
    if os.environ.get("BISECT_CLEAR", "1") == "1":      # drop the previous iteration's autograd graph (stale AccumulateGrad nodes)
        import gc
        m.fake_nhwc = m.fake_p2 = None
        m.loss_G_L1 = m.loss_G_GAN_PB = m.loss_G_GAN_PP = None
        m._fake_cats = None
        gc.collect()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    print("OK", stage, norm, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[1], sys.argv[2])
    else:
        for norm in ("batch",):
            for st in STAGES:
                r = subprocess.run([sys.executable, __file__, st, norm], capture_output=True, text=True, timeout=600)
                print(st, norm, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], flush=True)
                if r.returncode != 0:
                    print("   ", " | ".join(l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "File \"/root" in l)[-600:], flush=True)
