import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration, AdamWScale, train_step
cfg = FAT5Config(); cfg.fuse_norm_linear = True
m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
for clip in (None, 1.0):
    opt = AdamWScale(m.parameters(), lr=1e-3, weight_decay=0.0, kahan_sum=True, **({"max_grad_norm": clip} if clip else {}))
    for i in range(3):
        loss = m(ids, labels); loss.backward(); torch.cuda.synchronize(); print("bwd ok", clip, i, float(loss), flush=True)
        if "--torchclip" in sys.argv:
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0); torch.cuda.synchronize(); print("torch clip ok", flush=True)
        opt.step(); torch.cuda.synchronize(); print("step ok", flush=True)
        opt.zero_grad(set_to_none=True)
