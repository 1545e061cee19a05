"""Training driver with the reference's command line (src/main.py:10-70, flags of src/utils.py:51-127).

    python -m bsarec_amd.main --data_dir /path/to/data/ --data_name Beauty --lr 0.0005 --alpha 0.7 --c 5 \\
        --num_attention_heads 1 --train_name BSARec_Beauty

Same flow as the reference: read `<data_dir><data_name>.txt`, build the train / valid / test splits, train with
early stopping on validation NDCG@20 (patience epochs without improvement), reload the best parameters, report the
six test metrics.  Everything numeric runs in libbsarec_hip.so; batches come from the device-resident sample table.
"""
from __future__ import annotations

import argparse
import datetime
import logging
import os
import random
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

from . import data as D
from .model import MODEL_DICT
from .trainer import Trainer


def parse_args(argv=None):
    """The reference's flag names and defaults (src/utils.py:53-96); BSARec-specific --c / --alpha included."""
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", default="./data/", type=str)
    p.add_argument("--output_dir", default="output/", type=str)
    p.add_argument("--data_name", default="Beauty", type=str)
    p.add_argument("--do_eval", action="store_true")
    p.add_argument("--load_model", default=None, type=str)
    p.add_argument("--train_name", default=datetime.datetime.now().strftime('%b-%d-%Y_%H-%M-%S'), type=str)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--batch_size", default=256, type=int)
    p.add_argument("--epochs", default=200, type=int)
    p.add_argument("--no_cuda", action="store_true")
    p.add_argument("--log_freq", default=1, type=int)
    p.add_argument("--patience", default=10, type=int)
    p.add_argument("--num_workers", default=4, type=int)          # accepted, unused: batches are device-resident
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--weight_decay", default=0.0, type=float)
    p.add_argument("--adam_beta1", default=0.9, type=float)
    p.add_argument("--adam_beta2", default=0.999, type=float)
    p.add_argument("--gpu_id", default="0", type=str)
    p.add_argument("--model_type", default="BSARec", type=str)
    p.add_argument("--max_seq_length", default=50, type=int)
    p.add_argument("--hidden_size", default=64, type=int)
    p.add_argument("--num_hidden_layers", default=2, type=int)
    p.add_argument("--hidden_act", default="gelu", type=str)
    p.add_argument("--num_attention_heads", default=2, type=int)
    p.add_argument("--attention_probs_dropout_prob", default=0.5, type=float)
    p.add_argument("--hidden_dropout_prob", default=0.5, type=float)
    p.add_argument("--initializer_range", default=0.02, type=float)
    p.add_argument("--c", default=3, type=int)
    p.add_argument("--alpha", default=0.9, type=float)
    # DuoRec's flags (src/utils.py:106-111)
    p.add_argument("--tau", default=1.0, type=float)
    p.add_argument("--lmd", default=0.1, type=float)
    p.add_argument("--lmd_sem", default=0.1, type=float)
    p.add_argument("--ssl", default="us_x", type=str)
    p.add_argument("--sim", default="dot", type=str)
    return p.parse_args(argv)


class EarlyStopping:
    """src/utils.py:129-176: stop after `patience` validations without an improvement of the monitored score
    (NDCG@20); the best parameters are kept (in memory, and on disk when a path is given)."""

    def __init__(self, checkpoint_path, logger, patience=10):
        self.checkpoint_path, self.logger, self.patience = checkpoint_path, logger, patience
        self.counter, self.best_score, self.best_state, self.early_stop = 0, None, None, False

    def __call__(self, score, model):
        if self.best_score is None or np.any(score > self.best_score):
            self.best_score = score
            self.best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            if self.checkpoint_path:
                torch.save({k: v.cpu() for k, v in self.best_state.items()}, self.checkpoint_path)
            self.counter = 0
        else:
            self.counter += 1
            self.logger.info(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            self.early_stop = self.counter >= self.patience


def run(args, user_seq, logger=None, checkpoint_path=None):
    """Train + test on the given user sequences.  Returns (test scores, info string, epochs run, seconds)."""
    logger = logger or logging.getLogger("bsarec_amd")
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    max_item = max(max(s) for s in user_seq)
    args.item_size = max_item + 1                                   # src/main.py:23
    args.num_users = len(user_seq) + 1
    L, dev = args.max_seq_length, torch.device("cuda", torch.cuda.current_device())
    u, x, a = D.train_table(user_seq, L)
    train_dl = D.DeviceBatches(u, x, a, args.batch_size, dev, shuffle=True, seed=args.seed)
    eval_dl = D.DeviceBatches(*D.eval_table(user_seq, L, "valid"), args.batch_size, dev, shuffle=False)
    test_dl = D.DeviceBatches(*D.eval_table(user_seq, L, "test"), args.batch_size, dev, shuffle=False)
    n_users = len(user_seq)
    for split in ("valid", "test"):
        indptr, cols = D.seen_csr(user_seq, split)
        setattr(args, f"{split}_rating_matrix", sp.csr_matrix((np.ones(len(cols)), cols, indptr), shape=(n_users, args.item_size)))
    model = MODEL_DICT[args.model_type.lower()](args=args)
    if getattr(model, "needs_negatives", False):
        train_dl.enable_negatives(user_seq, args.item_size)
    if getattr(model, "needs_same_target", False):
        train_dl.enable_same_target()
    model.set_seed(args.seed)
    trainer = Trainer(model, train_dl, eval_dl, test_dl, args, logger)
    if args.do_eval:
        if args.load_model is None:
            logger.info("No model input!")
            return None
        trainer.load(os.path.join(args.output_dir, args.load_model + ".pt"))
        scores, info = trainer.test(0)
        return scores, info, 0, 0.0
    stopper = EarlyStopping(checkpoint_path, logger, patience=args.patience)
    t0 = time.time()
    epochs = 0
    for epoch in range(args.epochs):
        trainer.train(epoch)
        scores, _ = trainer.valid(epoch)
        stopper(np.array(scores[-1:]), trainer.model)               # monitors NDCG@20 (src/main.py:57)
        epochs = epoch + 1
        if stopper.early_stop:
            logger.info("Early stopping")
            break
    secs = time.time() - t0
    logger.info("---------------Test Score---------------")
    trainer.model.load_state_dict(stopper.best_state)
    scores, info = trainer.test(0)
    logger.info(args.train_name)
    logger.info(info)
    return scores, info, epochs, secs


def main(argv=None):
    args = parse_args(argv)
    os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu_id
    os.makedirs(args.output_dir, exist_ok=True)
    logger = logging.getLogger("bsarec_amd." + args.train_name)      # one log file per run (src/utils.py:9-28)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    fmt = logging.Formatter("%(asctime)s - %(message)s")
    for h in (logging.FileHandler(os.path.join(args.output_dir, args.train_name + ".log")), logging.StreamHandler(sys.stderr)):
        h.setFormatter(fmt)
        logger.addHandler(h)
    user_seq, _, _ = D.read_user_seqs(args.data_dir + args.data_name + ".txt")
    logger.info(str(args))
    return run(args, user_seq, logger, os.path.join(args.output_dir, args.train_name + ".pt"))


if __name__ == "__main__":
    main()
