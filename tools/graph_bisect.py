import os, sys
sys.path.insert(0, "/root/repo")
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
which = sys.argv[1]
cfg = FAT5Config(); cfg.crossentropy_inplace_backward = os.environ.get("INPLACE", "1") == "1"; cfg.num_layers = int(os.environ.get("NL", "1")); cfg.num_decoder_layers = int(os.environ.get("NL", "1"))
torch.manual_seed(0)
m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
def f():
    if which == "enc":
        m.encoder(ids).float().sum().backward()
    elif which == "dec":
        enc = torch.zeros(4, 1024, 768, device="cuda", dtype=torch.bfloat16)
        m.decoder(m._shift_right(labels), encoder_hidden_states=enc).float().sum().backward()
    elif which == "encdec":
        enc = m.encoder(ids)
        m.decoder(m._shift_right(labels), encoder_hidden_states=enc).float().sum().backward()
    elif which == "head":
        dec = torch.randn(4, 512, 768, device="cuda").bfloat16()
        loss = m.loss_fct(m.lm_head(dec), labels)
        loss.backward()
        return loss
    elif which == "torchonly":
        dec = torch.randn(4, 512, 768, device="cuda").bfloat16()
        loss = torch.nn.functional.cross_entropy(m.lm_head(dec).float().view(-1, 32768), labels.view(-1))
        loss.backward()
        return loss
    elif which == "dec_ret":
        enc = torch.zeros(4, 1024, 768, device="cuda", dtype=torch.bfloat16)
        loss = m.decoder(m._shift_right(labels), encoder_hidden_states=enc).float().mean()
        loss.backward()
        return loss
    elif which == "encdec_ret":
        enc = m.encoder(ids)
        loss = m.decoder(m._shift_right(labels), encoder_hidden_states=enc).float().mean()
        loss.backward()
        return loss
    elif which == "dechead_ret":
        enc = torch.zeros(4, 1024, 768, device="cuda", dtype=torch.bfloat16)
        dec = m.decoder(m._shift_right(labels), encoder_hidden_states=enc)
        loss = m.loss_fct(m.lm_head(dec), labels)
        loss.backward()
        return loss
    elif which == "enc_ret":
        loss = m.encoder(ids).float().mean()
        loss.backward()
        return loss
    elif which == "full":
        m(ids, labels).backward()
    elif which == "full_torchce":
        enc = m.encoder(ids)
        dec = m.decoder(m._shift_right(labels), encoder_hidden_states=enc)
        loss = torch.nn.functional.cross_entropy(m.lm_head(dec).float().view(-1, 32768), labels.view(-1))
        loss.backward()
        return loss
    elif which == "full_nohead":
        enc = m.encoder(ids)
        dec = m.decoder(m._shift_right(labels), encoder_hidden_states=enc)
        loss = m.lm_head(dec).float().mean()
        loss.backward()
        return loss
    elif which == "full_ret":
        loss = m(ids, labels)
        loss.backward()
        return loss
if os.environ.get("PRE") == "1":
    for _ in range(3):
        m.zero_grad(set_to_none=True); f()
    torch.cuda.synchronize()
if os.environ.get("PRE") == "2":
    for _ in range(3):
        m.zero_grad(set_to_none=True); f()
    f()
    torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m.zero_grad(set_to_none=True); f()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
m.zero_grad(set_to_none=True)
host = torch.zeros((), dtype=torch.float32).pin_memory()
with torch.cuda.graph(gr):
    kept = f()
    if os.environ.get("ITEM") == "2":
        host.copy_(kept.detach().float(), non_blocking=True)  # the read-back is a node of the graph
torch.cuda.synchronize()
if os.environ.get("ITEM") == "2":
    for i in range(5):
        gr.replay(); torch.cuda.synchronize(); print("host loss", float(host))
if os.environ.get("ITEM") == "1":
    gr.replay(); torch.cuda.synchronize()
    print("loss", (kept.clone().item() if os.environ.get("CLONE") == "1" else kept.item()))
for i in range(10):
    gr.replay()
torch.cuda.synchronize()
print(which, "OK")
