"""CPU, gloo, 2 ranks: tools/dp_launch.py wraps an UNCHANGED train script per rank -- sharded training loader, parameter broadcast,
one flat gradient all-reduce hooked into optimizer.step(), BatchNorm buffers averaged before evaluation.  Hermetic: the "reference"
is a small fake tree written to tmp_path with the reference's seams (train_semantics.py:4,47,110-119,186,320-323: `from
torch.utils.data import DataLoader`, `main(args)`, a shuffling train loader, `optim.AdamW(model.parameters())`, a lazily imported
`models.trainer.Trainer(model, optimizer, cfg, scheduler=...)(train_loader, val_loader)`)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import json, os, argparse
import torch
from torch.utils.data import DataLoader
import torch.optim as optim

class DS(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(5)
        self.x = torch.randn(n, 3, generator=g); self.y = torch.randn(n, 1, generator=g)
    def __len__(self): return len(self.x)
    def __getitem__(self, i): return i, self.x[i], self.y[i]

def main(args):
    cfg = json.load(open(args.cfg_path))
    train = DataLoader(dataset=DS(8), batch_size=4, shuffle=True, num_workers=0)
    val = DataLoader(DS(6), batch_size=1, shuffle=False, num_workers=0)
    torch.manual_seed(100 + int(os.environ.get("RANK", "0")))          # ranks start DIFFERENT: the launcher must broadcast
    model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 1))
    optimizer = optim.SGD(model.parameters(), lr=0.05)
    from models.trainer import Trainer
    trainer = Trainer(model, optimizer, cfg, scheduler=None, visualize=args.visualization, logging=args.with_logging)
    trainer(train, val)
"""

TRAINER = """
import json, os
import torch

class Trainer:
    def __init__(self, model, optimizer, cfg, scheduler=None, visualize=False, logging=False, test_mask=None):
        self.model, self.optimizer, self.cfg, self.logging = model, optimizer, cfg, logging
        self.seen = []
    def train_one_epoch(self, loader, epoch):
        self.model.train()
        for idx, x, y in loader:
            self.seen += idx.tolist()
            self.optimizer.zero_grad()
            loss = (self.model(x) - y).square().mean()
            loss.backward()
            self.optimizer.step()
    def test_one_epoch(self, loader, epoch):
        self.model.eval()
        n = 0
        for idx, x, y in loader:
            n += len(idx)
        self.val_seen = n
        return 0.5
    def __call__(self, train_loader, val_loader):
        for epoch in range(2):
            self.train_one_epoch(train_loader, epoch)
        self.test_one_epoch(val_loader, 1)
        rank = int(os.environ.get("RANK", "0"))
        sd = {k: v.tolist() for k, v in self.model.state_dict().items()}
        json.dump({"seen": self.seen, "val_seen": self.val_seen, "logging": bool(self.logging), "sd": sd},
                  open(os.path.join(self.cfg["out_dir"], f"rank{rank}.json"), "w"))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_launch_wraps_an_unchanged_train_script(tmp_path):
    src = tmp_path / "src"
    (src / "models").mkdir(parents=True)
    (src / "train_like.py").write_text(textwrap.dedent(SCRIPT))
    (src / "models" / "trainer.py").write_text(textwrap.dedent(TRAINER))
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps({"out_dir": str(tmp_path)}))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tools", "dp_launch.py"), "--script", str(src / "train_like.py"), "--backend", "gloo",
           "--cfg_path", str(cfg), "--mode", "train", "--with_logging", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    # seam 1: per epoch the two ranks see disjoint halves of the 8 training samples; a new permutation each epoch; validation whole
    for e in range(2):
        a, b = out[0]["seen"][4 * e:4 * e + 4], out[1]["seen"][4 * e:4 * e + 4]
        assert sorted(a + b) == list(range(8))
    assert out[0]["seen"][:4] != out[0]["seen"][4:] and out[0]["val_seen"] == out[1]["val_seen"] == 6
    # seams 2 + 3: different initial seeds, identical final parameters AND BatchNorm statistics on both ranks
    for k, v in out[0]["sd"].items():
        w = out[1]["sd"][k]
        assert v == w or abs(sum(map(float, _flat(v))) - sum(map(float, _flat(w)))) < 1e-6, k
    assert out[0]["logging"] is True and out[1]["logging"] is False            # only rank 0 logs / checkpoints


def _flat(v):
    if isinstance(v, list):
        for x in v:
            yield from _flat(x)
    else:
        yield v


def test_dp_launch_failing_rank_exits_without_joining_collectives(tmp_path):
    """Rank 1 raises in its first training step while rank 0 is inside the gradient all-reduce: the failing rank must leave with a
    non-zero code WITHOUT a barrier (which would pair with the peer's all-reduce), so the job fails fast instead of hanging."""
    src = tmp_path / "src"
    (src / "models").mkdir(parents=True)
    (src / "train_like.py").write_text(textwrap.dedent(SCRIPT))
    failing = TRAINER.replace("            self.optimizer.zero_grad()",
                              "            if int(os.environ.get('RANK', '0')) == 1:\n                raise KeyError('unmapped label 77')\n"
                              "            self.optimizer.zero_grad()")
    assert failing != TRAINER
    (src / "models" / "trainer.py").write_text(textwrap.dedent(failing))
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps({"out_dir": str(tmp_path)}))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tools", "dp_launch.py"), "--script", str(src / "train_like.py"), "--backend", "gloo",
           "--cfg_path", str(cfg), "--mode", "train"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)      # a hang would hit the timeout
    assert r.returncode != 0
    assert "unmapped label 77" in r.stderr and "exiting without joining further collectives" in r.stderr
    assert not (tmp_path / "rank0.json").exists() and not (tmp_path / "rank1.json").exists()
