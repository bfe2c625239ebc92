import sys, os, torch
sys.path.insert(0, "/root/repo")
from hipt_abmil_atec23_amd import HIPT_4K, synth
DEV = "cuda:0"
m = HIPT_4K(None, None, DEV, DEV)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(DEV).set_compute_dtype("bf16")
n = 4
f32 = torch.cat([synth.hash_uniform_torch((1, 3, 1024, 1024), 300 + i, device=DEV) for i in range(n)])
u8 = ((f32 * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
u8p = ((f32 * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).contiguous()
junk = torch.randint(0, 255, (64 << 20,), dtype=torch.uint8, device=DEV)
for streams in (1, 2):
    m.streams = streams
    for name, x in (("f32", f32), ("u8hwc", u8), ("u8chw", u8p)):
        for nreg in (1, 2, 3, 4):
            ref = [m(x[i:i + 1].clone()).clone() for i in range(nreg)]
            bad = 0
            for rep in range(30):
                # a fresh buffer each time (different neighbours in memory), sometimes embedded in a larger one
                if rep % 2:
                    big = torch.randint(0, 255, (nreg + 1,) + tuple(x.shape[1:]), device=DEV).to(x.dtype) if x.dtype == torch.uint8 else torch.randn((nreg + 1,) + tuple(x.shape[1:]), device=DEV)
                    big[:nreg] = x[:nreg]
                    xin = big[:nreg]
                else:
                    xin = x[:nreg].clone()
                o = m(xin)
                torch.cuda.synchronize()
                eq = [bool(torch.equal(o[i], ref[i][0])) for i in range(nreg)]
                bad += 0 if all(eq) else 1
                if not all(eq) and bad <= 2:
                    print("   mismatch", name, "streams", streams, "nreg", nreg, "rep", rep, eq, flush=True)
            print(name, "streams", streams, "nreg", nreg, "bad", bad, "/ 30", flush=True)
