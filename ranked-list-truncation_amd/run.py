#!/usr/bin/env python3
"""Training loop of the truncation models on the HIP hot path - the mirror of the reference's
run.py `Trainer` (:26-240) and `main` (:301-369) for the five in-scope models.

Same flow per step (run.py:120-151): zero_grad -> model(X) -> criterion -> backward -> Adam ->
k = argmax+1 -> F1/DCG; same per-epoch bookkeeping (unweighted means over steps, best / best-5 test F1,
state_dict checkpoint on best test F1, run.py:153-232) and the same log lines.  What differs:
  * model / criterion / metrics / optimizer run through librlt_hip.so; inputs are fed by the
    pinned-memory loader in dataloader/ (the reference's pickle layout);
  * F1/DCG are evaluated on the device (no (B,S) round trip per step); one host sync per step for
    the three logged scalars;
  * launched under torch.distributed.run it is data-parallel: all ranks draw the same batch permutation (one seed
    broadcast from rank 0), every rank takes its contiguous shard of each batch (an exact partition, ragged tails
    included: shards are weighted by their list counts in the loss, the gradient and the logged means), one RCCL
    all-reduce of the flat gradient bucket per step (rlt_hip/parallel.py);
  * the scalars the reference sends to tensorboardX (run.py:146,154-156,196-198) go, under the same tags, to
    `<tensorboard-dir>/scalars.jsonl` (and to a SummaryWriter when tensorboard is installed);
  * not carried over: matplotlib plots and the hyper-parameter random search.
"""
import argparse
import configparser
import json
import logging
import math
import os
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
logging.basicConfig(level=logging.INFO)

from dataloader import at_dataloader, cp_dataloader, mc_dataloader, write_synthetic_robust04  # noqa: E402
from models import AttnCut, BiCut, Choopy, MMOECut, MOECut, MtAttnCut, MtChoopy, PLECut  # noqa: E402
from utils import losses  # noqa: E402
from utils.metrics import Metric  # noqa: E402
from rlt_hip import ops  # noqa: E402
from rlt_hip.parallel import FORCE_COLLECTIVES, FlatModel, FusedAdam, shard_bounds  # noqa: E402


class ScalarLog:
    """add_scalar(tag, value, step) of the reference's tensorboardX writer (run.py:111): one JSON line per scalar in
    <dir>/scalars.jsonl, mirrored into torch.utils.tensorboard when that package is importable."""

    def __init__(self, log_dir):
        self.file = self.tb = None
        if not log_dir:
            return
        os.makedirs(log_dir, exist_ok=True)
        self.file = open(os.path.join(log_dir, "scalars.jsonl"), "w")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(log_dir=log_dir)
        except Exception:                      # tensorboard is not part of this image
            self.tb = None

    def add_scalar(self, tag, value, step):
        if self.file is not None:
            self.file.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        if self.tb is not None:
            self.tb.add_scalar(tag, value, step)

    def close(self):
        if self.file is not None:
            self.file.close()
            self.file = None
        if self.tb is not None:
            self.tb.close()


class Trainer:
    def __init__(self, args):
        self.args = args
        self.seq_len = 300 if args.retrieve_data == 'robust04' else 40
        self.model_name = args.model_name
        self.epochs, self.batch_size = args.epochs, args.batch_size
        self.model_persist, self.save_path, self.model_path = args.model_persist, args.save_path, args.model_path
        self.best_test_f1, self.best_test_dcg = -float('inf'), -float('inf')
        self.best_epoch = None                  # epoch whose weights were checkpointed (None: no test epoch has run)
        self.best5_f1 = self.best5_dcg = None
        self.f1_record, self.dcg_record = [], []
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if not torch.cuda.is_available():
            raise RuntimeError("run.py trains on the GPU through librlt_hip.so; there is no CPU fallback")
        # RLT_RUN_DEVICE / RLT_DIST_BACKEND exist only to rehearse the data-parallel loop on a one-GPU box (several ranks
        # on cuda:0 over gloo, as bench.py's RLT_BENCH_DEVICE does); real runs use one GPU per rank over RCCL
        self.device = torch.device("cuda", int(os.environ.get("RLT_RUN_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        torch.cuda.set_device(self.device)

        name = self.model_name
        if name in ('choopy', 'mtchoopy'):
            loader = cp_dataloader
        elif name in ('mmoecut', 'moecut', 'mtple') and args.retrieve_data != 'robust04':
            loader = mc_dataloader                                             # run.py:87-88: MQ2007 multi-task statistics
        else:
            loader = at_dataloader
        self.train_loader, self.test_loader, data = loader(args.retrieve_data, args.dataset_name, args.batch_size,
                                                           device=self.device, base=args.dataset_base, seed=args.seed)
        # the reference hard-codes 3 / 25 / 47 input features and 300 / 40 positions (run.py:34,60,70,86); here both
        # come from the files, and a mismatch with the reference's numbers is reported instead of mis-striding the LSTM
        feat = data.n_features
        if len(data.lengths) == 1:
            self.seq_len = data.lengths[0]
        if name not in ('bicut', 'attncut', 'mtattncut') and (len(data.lengths) != 1 or data.test_lengths != data.lengths):
            # Choopy's position encoding and the MMOE gates are sized by seq_len: a test list of another length would
            # only fail inside the model at the first test epoch
            raise ValueError(f"{name} is built for ONE list length; the data holds train lengths {data.lengths} and test "
                             f"lengths {data.test_lengths} (only the BiLSTM models bicut / attncut / mtattncut take "
                             "length-bucketed batches)")
        if name == 'bicut':                                                    # run.py:59-64
            self.model = BiCut(input_size=feat, dropout=args.dropout)
            self.criterion = losses.BiCutLoss(metric=args.criterion)
        elif name == 'choopy':                                                 # run.py:65-68
            self.model = Choopy(seq_len=self.seq_len, dropout=args.dropout)
            self.criterion = losses.ChoopyLoss(metric=args.criterion)
        elif name == 'attncut':                                                # run.py:69-75
            self.model = AttnCut(input_size=feat, dropout=args.dropout)
            self.criterion = losses.DivLoss(metric=args.criterion, div_type=args.div_type,
                                            augmented=args.augmented_reward)
        elif name == 'mtchoopy':                                               # run.py:76-79
            self.model = MtChoopy(seq_len=self.seq_len, num_tasks=args.num_tasks, dropout=args.dropout)
            self.criterion = losses.MtCutLoss(metric=args.criterion, rerank_weight=args.rerank_weight,
                                              classi_weight=args.class_weight, num_tasks=args.num_tasks)
        elif name == 'mtattncut':                                              # run.py:80-84
            self.model = MtAttnCut(input_size=feat, num_tasks=args.num_tasks, dropout=args.dropout)
            self.criterion = losses.MtCutLoss(metric=args.criterion, rerank_weight=args.rerank_weight,
                                              classi_weight=args.class_weight, num_tasks=args.num_tasks)
        elif name == 'mmoecut':                                                # run.py:85-90 (num_experts exposed)
            self.model = MMOECut(seq_len=self.seq_len, num_tasks=args.num_tasks, input_size=feat,
                                 dropout=args.dropout, num_experts=args.num_experts)
            self.criterion = losses.MtCutLoss(metric=args.criterion, num_tasks=args.num_tasks)
        elif name == 'moecut':                                                 # run.py:91-96
            self.model = MOECut(seq_len=self.seq_len, num_tasks=args.num_tasks, input_size=feat, dropout=args.dropout)
            self.criterion = losses.MtCutLoss(metric=args.criterion, num_tasks=args.num_tasks)
        elif name == 'mtple':                                                  # run.py:97-102
            self.model = PLECut(seq_len=self.seq_len, input_size=feat, dropout=args.dropout, num_experts=3)
            self.criterion = losses.MtCutLoss(metric=args.criterion, num_tasks=args.num_tasks)
        else:
            raise ValueError(f"model {name!r} is outside the HIP hot path "
                             "(bicut, choopy, attncut, mtchoopy, mtattncut, mmoecut, moecut, mtple)")
        self.multi_task = name in ('mtchoopy', 'mtattncut', 'mmoecut', 'moecut', 'mtple')   # reference: `'m' in model_name`
        self.model = self.model.to(self.device)
        if args.ft and self.model_path and os.path.exists(self.model_path):
            self.load_model()
        self.flat = FlatModel(self.model)
        self.flat.broadcast_params()
        self.optimizer = FusedAdam(self.flat, lr=args.lr, weight_decay=args.weight_decay)   # run.py:104
        ops.set_seed_stream(self.rank)          # same torch seed on every rank, different dropout masks
        self.writer = ScalarLog(getattr(args, "tensorboard_dir", None) if self.rank == 0 else None)   # run.py:111
        self.history = []                       # per epoch: train / test (loss, f1, dcg) means

    # ------------------------------------------------------------------------------------------
    def _step(self, X, y, train):
        """One batch.  Data-parallel: every rank holds the SAME batch (shared permutation) and works on its contiguous
        shard; shards partition the batch exactly (a ragged tail gives unequal, possibly empty shards), so loss,
        gradient and logged means are weighted by the shards' list counts - what comes out is the batch mean of the
        shard-wise reference computation."""
        n_all = X.shape[0]
        lo, hi = shard_bounds(n_all, self.rank, self.world)
        n_own = hi - lo
        stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        if n_own > 0:
            Xs, ys = (X, y) if self.world == 1 else (X[lo:hi], y[lo:hi])
            output = self.model(Xs)
            # loss and cut metrics (run.py:126,131-145) out of one pass over p and the labels; BiCut's (B,S,2) output goes
            # through its own criterion and Metric.evaluate's cut rule
            loss, _k, f1, dcg = Metric.step(self.criterion, output, ys)
            if train:
                # AVG all-reduce of sum_r (n_r * world / n) * grad_r / world = sum_r (n_r / n) * grad_r
                (loss if n_own * self.world == n_all else loss * (n_own * self.world / n_all)).backward()
            stats = torch.stack([loss.detach().double(), f1, dcg, torch.ones((), dtype=torch.float64, device=self.device)]) * n_own
        if train:
            self.flat.all_reduce_grads()                        # an empty shard contributes its zeroed bucket
            self.optimizer.step()
        if self.world > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        vals = stats.tolist()                                   # the step's only host sync
        return [v / vals[3] for v in vals[:3]]

    def train_epoch(self, epoch):
        start = time.time()
        tot, step, num_itr = [0.0, 0.0, 0.0], 0, len(self.train_loader)
        logging.info('-' * 100)
        for X, y in self.train_loader:
            self.model.train()
            self.optimizer.zero_grad()
            vals = self._step(X, y, True)
            self.writer.add_scalar('train/loss_step', vals[0], step + num_itr * epoch)   # run.py:146
            tot = [a + b for a, b in zip(tot, vals)]
            step += 1
        loss, f1, dcg = [v / step for v in tot]                 # unweighted over steps, run.py:153
        self.writer.add_scalar('train/loss_epoch', loss, epoch)
        self.writer.add_scalar('train/F1_epoch', f1, epoch)
        self.writer.add_scalar('train/DCG_epoch', dcg, epoch)
        self.history.append({"epoch": epoch, "train": (loss, f1, dcg)})
        if self.rank == 0:
            logging.info('\nEpoch: {} | Epoch Time: {:.2f} s'.format(epoch, time.time() - start))
            logging.info('\tTrain: loss = {} | f1 = {:.6f} | dcg = {:.6f}\n'.format(loss, f1, dcg))
        return loss, f1, dcg

    def test(self, epoch):
        tot, step = [0.0, 0.0, 0.0], 0
        for X, y in self.test_loader:
            self.model.eval()
            with torch.no_grad():
                vals = self._step(X, y, False)
            tot = [a + b for a, b in zip(tot, vals)]
            step += 1
        loss, f1, dcg = [v / step for v in tot]                 # run.py:195
        self.writer.add_scalar('test/loss_epoch', loss, epoch)
        self.writer.add_scalar('test/F1_epoch', f1, epoch)
        self.writer.add_scalar('test/DCG_epoch', dcg, epoch)
        if self.history and self.history[-1]["epoch"] == epoch:
            self.history[-1]["test"] = (loss, f1, dcg)
        self.f1_record.append(f1)
        self.dcg_record.append(dcg)
        if self.rank == 0:
            logging.info('\tTest: loss = {} | f1 = {:.6f} | dcg = {:.6f}\n'.format(loss, f1, dcg))
        if f1 > self.best_test_f1:                              # run.py:203-206
            self.best_test_f1, self.best_epoch = f1, epoch
            if self.model_persist and self.rank == 0:
                self.save_model()
        if dcg > self.best_test_dcg:
            self.best_test_dcg = dcg
        return loss, f1, dcg

    def save_model(self):
        os.makedirs(self.save_path, exist_ok=True)
        # clone: parameters are views of the flat bucket
        state = {k: v.detach().clone().cpu() for k, v in self.model.state_dict().items()}
        torch.save(state, os.path.join(self.save_path, '{}.pkl'.format(self.model_name)))
        logging.info('The best model has beed updated and saved in {}\n'.format(self.save_path))

    def load_model(self):
        self.model.load_state_dict(torch.load(self.model_path, map_location=self.device))
        logging.info('The best model has beed loaded from {}\n'.format(self.model_path))

    def run(self):
        if self.rank == 0:
            logging.info('\nTrain the {} model: \n'.format(self.model_name))
        for epoch in range(self.epochs):
            self.train_epoch(epoch)
            self.test(epoch)
        top = sorted(self.f1_record, reverse=True)[:5]
        topd = sorted(self.dcg_record, reverse=True)[:5]
        best5_f1, best5_dcg = sum(top) / 5, sum(topd) / 5       # run.py:229-230 divides by 5 regardless
        self.best5_f1, self.best5_dcg = best5_f1, best5_dcg
        self.writer.close()
        if self.rank == 0:
            logging.info('the best metric of this model: f1: {} | dcg: {}'.format(self.best_test_f1, self.best_test_dcg))
            logging.info('the best-5 metric of this model: f1: {} | dcg: {}'.format(best5_f1, best5_dcg))
        return self.best_test_f1, self.best_test_dcg


def build_parser():
    p = argparse.ArgumentParser(description="Truncation Model Trainer Args (HIP hot path)")
    p.add_argument('--retrieve-data', type=str, default='robust04')
    p.add_argument('--dataset-name', type=str, default='drmm_tks')
    p.add_argument('--batch-size', type=int, default=63)
    p.add_argument('--model-name', type=str, default='mmoecut')
    p.add_argument('--augmented-reward', type=int, default=1)
    p.add_argument('--div-type', type=str, default='js')
    p.add_argument('--criterion', type=str, default='dcg')
    p.add_argument('--model-path', type=str, default=None)
    p.add_argument('--ft', type=int, default=0)
    p.add_argument('--model-persist', type=int, default=0)
    p.add_argument('--save-path', type=str, default=os.path.join(HERE, 'best_model'))
    p.add_argument('--epochs', type=int, default=80)
    p.add_argument('--lr', type=float, default=3e-5)
    p.add_argument('--weight-decay', type=float, default=0.005)
    p.add_argument('--dropout', type=float, default=0.1)
    p.add_argument('--num-tasks', type=float, default=3)       # 2.1: class + cut | 2.2: rerank + cut
    p.add_argument('--rerank-weight', type=float, default=0.3)
    p.add_argument('--class-weight', type=float, default=0.4)
    # additions
    p.add_argument('--num-experts', type=int, default=3, help="MMOECut experts (the reference hard-codes 3, run.py:89)")
    p.add_argument('--dataset-base', type=str, default=None, help="directory holding <retrieve_data>/*.pkl")
    p.add_argument('--synthetic', type=int, default=0, help="write a robust04-shaped synthetic set into --dataset-base first")
    p.add_argument('--use-conf', type=int, default=1, help="override lr/batch/dropout/wd/task weights from hyper_parameter_<dataset>.conf")
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--history-json', type=str, default=None, help="rank 0 writes the per-epoch train / test means and the best / best-5 figures here")
    p.add_argument('--param-dump-dir', type=str, default=None,
                   help="every rank saves its final flat parameter bucket as flat_param_rank<r>.npy here (data-parallel checks: the "
                        "replicas must stay bitwise identical)")
    p.add_argument('--tensorboard-dir', type=str, default=os.path.join(HERE, 'Tensorboard_summary', 'Truncation'),
                   help="scalars.jsonl (+ tensorboard event files when tensorboard is installed); '' disables")
    return p


def apply_conf(args):
    """run.py:338-347: the .conf section of the model overrides the CLI values."""
    conf = configparser.ConfigParser()
    path = os.path.join(HERE, 'hyper_parameter_{}.conf'.format(args.dataset_name))
    sec = '{}_conf'.format(args.model_name)
    if not conf.read(path) or not conf.has_section(sec):
        # the reference dies with configparser.NoSectionError here (run.py:340); running on silently with the CLI
        # defaults would train a different model than the user asked for
        raise configparser.NoSectionError(f"{sec} (in {path}; pass --use-conf 0 to train with the command-line values)")
    args.lr = conf.getfloat(sec, 'lr')
    if args.retrieve_data == 'robust04':
        args.batch_size = conf.getint(sec, 'batch_size')
    args.dropout = conf.getfloat(sec, 'dropout')
    args.weight_decay = conf.getfloat(sec, 'weight_decay')
    if 'm' in args.model_name:
        args.rerank_weight = conf.getfloat(sec, 'rerank_weight')
        args.class_weight = conf.getfloat(sec, 'class_weight')
    return args


def main(argv=None):
    args = build_parser().parse_args(argv)
    # RLT_FORCE_DIST=1: initialise the process group (and run the step's collectives) even with one rank - the RCCL
    # rehearsal on a one-GPU box (rlt_hip/parallel.py)
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or (FORCE_COLLECTIVES and "RANK" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RLT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.use_conf:
        args = apply_conf(args)
    if args.model_path is None:
        args.model_path = os.path.join(args.save_path, '{}.pkl'.format(args.model_name))
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if args.synthetic:
        if not args.dataset_base:
            raise SystemExit("--synthetic needs --dataset-base")
        if (not dist.is_initialized()) or dist.get_rank() == 0:
            write_synthetic_robust04(args.dataset_base, args.retrieve_data, args.dataset_name)
        if dist.is_initialized():
            dist.barrier()
    logging.info('{}'.format(vars(args)))
    trainer = Trainer(args)
    result = trainer.run()
    if args.history_json and trainer.rank == 0:
        with open(args.history_json, "w") as f:
            fin = lambda v: v if v is not None and math.isfinite(v) else None     # -inf (no test epoch) is not valid JSON
            json.dump({"history": trainer.history, "best_f1": fin(trainer.best_test_f1), "best_dcg": fin(trainer.best_test_dcg),
                       "best5_f1": fin(trainer.best5_f1), "best5_dcg": fin(trainer.best5_dcg), "best_epoch": trainer.best_epoch,
                       "world": trainer.world}, f)
    if args.param_dump_dir:
        import numpy as np
        os.makedirs(args.param_dump_dir, exist_ok=True)
        np.save(os.path.join(args.param_dump_dir, f"flat_param_rank{trainer.rank}.npy"), trainer.flat.flat_param.detach().cpu().numpy())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
