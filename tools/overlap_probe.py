"""What would running the two halves of a batch on two streams buy?  Emulated from Python: two fused steps on half the items each, on two
torch streams, against one fused step on the whole batch (C3: 16 sets -> 8 + 8, C4: 2 -> 1 + 1, C2: 32 -> 16 + 16)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unipre3d_amd import synthetic
from unipre3d_amd.fused import backward_unit, render_loss_fused
dev = torch.device("cuda")
t_pre = time.time()
x = torch.randn(4096, 4096, device=dev)
while time.time() - t_pre < 1.5: x = x @ x * 1e-4      # clocks up
for name in sys.argv[1:] or ["C2", "C3", "C4"]:
    c = synthetic.CONFIGS[name]
    b = synthetic.make_batch(c["B"], c["P"], c["V"], c["H"], c["W"], level=c["level"], seed=42).to(dev)
    kind = "focal_l2" if c["level"] == "object" else "l2"
    def mk(lo, hi):
        h = b.raw[lo:hi].permute(0, 2, 1).contiguous().requires_grad_(True)
        args = (b.center[lo:hi], b.world_view[lo:hi], b.full_proj[lo:hi], b.camera_center[lo:hi], b.gt[lo:hi].contiguous(), b.bg, b.fov_deg, c["H"], c["W"])
        def step():
            h.grad = None
            loss, _, _ = render_loss_fused(h, *args, level=c["level"], offset_scale=b.offset_scale, loss_kind=kind, return_images=False)
            backward_unit(loss)
        return step
    whole, halfA, halfB = mk(0, c["B"]), mk(0, c["B"] // 2), mk(c["B"] // 2, c["B"])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def both():
        with torch.cuda.stream(s1): halfA()
        with torch.cuda.stream(s2): halfB()
    def both_sync():        # with the fork / join a single call would need: halves start together and the caller waits for both
        ev = torch.cuda.current_stream().record_event()
        s1.wait_event(ev); s2.wait_event(ev)
        both()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    for label, fn in (("one call", whole), ("two halves, two streams, joined per step", both_sync), ("half A alone", halfA)):
        for _ in range(30): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize()
        print("%s %-44s %.1f us/step" % (name, label, 1e6 * (time.perf_counter() - t0) / 200))
