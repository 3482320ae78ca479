import sys, time, torch
sys.path.insert(0, '/root/repo')
from visitron_amd.config import BertConfig
from visitron_amd.modeling import PreTrainOscar
from visitron_amd.synth import make_batch
from visitron_amd.training import PretrainEngine
dev = torch.device("cuda:0")
cfg = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
torch.manual_seed(0)
m = PreTrainOscar(cfg).to(dev).train()
eng = PretrainEngine(m, lr=5e-5)
batches = [make_batch(cfg, 256, 128, 100, seed=100 + i, device=dev) for i in range(6)]
for i, b in enumerate(batches):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.train_step(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("step %d rows %d loss %.4f  %.1f ms" % (i, eng.last_rows, float(out[0]), dt))
