#!/usr/bin/env python3
"""CLI counterpart of the reference's `python train.py` (train.py:396-737) on the MI355X path: the same argument names for
what is built (dataset windows, GenieConfig json, AdamW + decay grouping, linear / custom_cosine schedules, gradient
accumulation, clipping, periodic teacher-forced eval, `save_pretrained` checkpoints with the optimizer state, resume).
Not built: accelerate/wandb logging, torch.compile, the Llama baseline, MuAdamW (--mu_transfer).

  python tools/train.py --genie_config genie/configs/magvit_n32_h8_d256.json --train_data_dir data/train_v1.1 \\
      --val_data_dir data/val_v1.1 --output_dir out --per_device_train_batch_size 8 --max_train_steps 1000
  python tools/train.py --synthetic 64 --output_dir /tmp/out --max_train_steps 5        # offline smoke: synthetic clips
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py ...   # data-parallel (RCCL)
"""
import argparse
import importlib
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="Train a GENIE spatio-temporal MaskGIT model (MI355X path).")
    p.add_argument("--train_data_dir", type=str, default="data/train_v1.1")
    p.add_argument("--val_data_dir", type=str, default="data/val_v1.1")
    p.add_argument("--window_size", type=int, default=16)
    p.add_argument("--stride", type=int, default=15)
    p.add_argument("--filter_overlaps", action="store_true")
    p.add_argument("--genie_config", type=str, help="GenieConfig json")
    p.add_argument("--warmstart_path", type=str, default=None)
    p.add_argument("--resume_from_checkpoint", type=str, default=None,
                   help="a step_N / final_checkpt directory written by this script (weights + trainer_state.pt)")
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--per_device_train_batch_size", type=int, default=4)
    p.add_argument("--per_device_eval_batch_size", type=int, default=4)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--num_train_epochs", type=int, default=1)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--max_eval_steps", type=int, default=int(1e10))
    p.add_argument("--eval_every_n_steps", type=int, default=1000)
    p.add_argument("--lr_scheduler_type", type=str, default="linear", choices=["linear", "constant", "custom_cosine"])
    p.add_argument("--num_warmup_steps", type=int, default=0)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--adam_beta_1", type=float, default=0.9)
    p.add_argument("--adam_beta_2", type=float, default=0.999)
    p.add_argument("--adam_eps", type=float, default=1e-8)
    p.add_argument("--checkpointing_steps", type=str, default="1000")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--mu_transfer", action="store_true")
    p.add_argument("--precision", choices=["exact", "f16x3", "bf16"], default="bf16",
                   help="bf16 = what the reference computes under --mixed_precision bf16; f16x3 keeps f32-class gradients")
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic clips (no dataset on disk)")
    p.add_argument("--model", choices=["c138", "c35", "tiny"], default="c35", help="shape when no --genie_config is given")
    return p.parse_args()


def main():
    args = parse_args()
    if args.mu_transfer:
        raise NotImplementedError("--mu_transfer needs the un-vendored mup fork (MuAdamW, set_base_shapes): not built")
    P = lambda n: importlib.import_module("1xgpt_amd." + n)  # noqa: E731
    dist_mod, cfgmod, synth, datamod, trainmod = P("distributed"), P("config"), P("synthetic"), P("data"), P("train")
    STMaskGIT = P("st_mask_git").STMaskGIT
    rank, world, local_rank = dist_mod.init_distributed()
    dev = torch.device("cuda", dist_mod.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    if args.seed is not None:
        import random
        torch.manual_seed(args.seed)  # rank-independent while the model is built (the per-rank offset follows below)
        random.seed(args.seed)  # the collator's branch draws are host-side: keep the ranks in step

    # ---- data (train.py:421-436)
    if args.synthetic:
        cfg = (cfgmod.GenieConfig.from_pretrained(args.genie_config) if args.genie_config else
               {"c138": cfgmod.c138, "c35": cfgmod.c35,
                "tiny": lambda: cfgmod.GenieConfig(num_layers=2, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2,
                                                   qk_norm=False, num_prompt_frames=2)}[args.model]())
        train_clips = torch.from_numpy(synth.make_clips(args.synthetic, cfg, seed=1))
        eval_clips = train_clips[: max(1, args.synthetic // 8)]
        get_train = lambda idx: train_clips[idx]  # noqa: E731
        get_eval = lambda idx: eval_clips[idx]  # noqa: E731
        n_train, n_eval = len(train_clips), len(eval_clips)
    else:
        tds = datamod.RawTokenDataset(args.train_data_dir, window_size=args.window_size, stride=args.stride,
                                      filter_overlaps=args.filter_overlaps)
        eds = datamod.RawTokenDataset(args.val_data_dir, window_size=args.window_size, stride=args.stride,
                                      filter_overlaps=True)
        assert all(tds.metadata[k] == eds.metadata[k] for k in ("s", "vocab_size", "hz"))
        cfg = cfgmod.GenieConfig.from_pretrained(args.genie_config)
        cfg.image_vocab_size, cfg.T, cfg.S = tds.metadata["vocab_size"], args.window_size, tds.metadata["s"] ** 2
        cfg.__post_init__()
        get_train, get_eval, n_train, n_eval = tds.batch, eds.batch, len(tds), len(eds)

    load_from = args.resume_from_checkpoint or args.warmstart_path
    model = (STMaskGIT.from_pretrained(load_from, precision=args.precision) if load_from
             else STMaskGIT(cfg, precision=args.precision))
    if not load_from and args.mu_transfer:
        # reference train.py:420-424: init_weights() only on the muP path; otherwise PyTorch's default initialisation
        # of nn.Linear / nn.Embedding stays (kept here too, so from-scratch dynamics follow the reference recipe)
        model.init_weights()
    model = model.to(dev)
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)  # collator draws differ per rank; the weights above do not

    B, accum = args.per_device_train_batch_size, args.gradient_accumulation_steps
    micro_per_epoch = n_train // (B * world)
    eb_ = args.per_device_eval_batch_size
    if micro_per_epoch < 1:
        sys.exit(f"train.py: {n_train} training windows < per_device_train_batch_size {B} x world {world}: no full batch")
    if n_eval < eb_ * world:
        sys.exit(f"train.py: {n_eval} eval windows < per_device_eval_batch_size {eb_} x world {world}: no full eval batch")
    updates_per_epoch = max(1, micro_per_epoch // accum)
    max_steps = args.max_train_steps or args.num_train_epochs * updates_per_epoch
    warm = args.num_warmup_steps
    lr_lambda = {"linear": trainmod.lr_factor_linear(warm, max_steps), "constant": lambda s: 1.0,
                 "custom_cosine": trainmod.lr_factor_custom_cosine(warm, max_steps)}[args.lr_scheduler_type]
    tr = trainmod.GenieTrainer(model, lr=args.learning_rate, betas=(args.adam_beta_1, args.adam_beta_2), eps=args.adam_eps,
                               weight_decay=args.weight_decay, max_grad_norm=args.max_grad_norm,
                               gradient_accumulation_steps=accum, lr_lambda=lr_lambda)
    if args.resume_from_checkpoint:  # optimizer moments, step counters (train.py:560-590)
        tr.load_state_dict(torch.load(os.path.join(args.resume_from_checkpoint, "trainer_state.pt"), map_location=dev))
    n_params = sum(p.numel() for p in model.parameters())
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        print(f"params {n_params / 1e6:.1f} M, {n_train} train windows, batch {B} x {accum} x {world}, "
              f"{max_steps} update steps, precision {args.precision}", flush=True)

    def evaluate():
        """Teacher-forced loss / accuracy on collated eval batches (train.py:665-697)."""
        sums = torch.zeros(3, dtype=torch.float64, device=dev)
        eb = args.per_device_eval_batch_size
        for k, s0 in enumerate(range(rank * eb, n_eval - eb * world + 1, eb * world)):
            batch = datamod.maskgit_collate(get_eval(range(s0, s0 + eb)).to(dev), cfg)
            out = model(batch["input_ids"], batch["labels"])
            sums += torch.stack([out.loss.double() * eb, out.acc.double() * eb, torch.tensor(float(eb), device=dev).double()])
            if k + 1 >= args.max_eval_steps:
                break
        if world > 1:
            torch.distributed.all_reduce(sums)
        return (sums[0] / sums[2]).item(), (sums[1] / sums[2]).item()

    ckpt_every = int(args.checkpointing_steps) if args.checkpointing_steps.isdigit() else None
    completed, t0 = tr.completed_steps, time.time()
    loss_info = torch.zeros(2, dtype=torch.float64, device=dev)
    consumed = completed * accum  # micro-batches already trained on (resume): skip them in the data order
    for epoch in range(consumed // micro_per_epoch, 10 ** 9):
        g = torch.Generator().manual_seed((args.seed or 0) + epoch)
        perm = torch.randperm(n_train, generator=g)  # same permutation on every rank; rank r takes its slice
        for m in range(consumed % micro_per_epoch if epoch == consumed // micro_per_epoch else 0, micro_per_epoch):
            idx = perm[(m * world + rank) * B:(m * world + rank + 1) * B].tolist()
            batch = datamod.maskgit_collate(get_train(idx).to(dev), cfg)
            out = tr.train_step(batch)
            loss_info += torch.stack([out["loss"] * B, torch.tensor(float(B), device=dev, dtype=torch.float64)])
            if "lr" not in out:
                continue
            completed += 1
            if world > 1:
                torch.distributed.all_reduce(loss_info)
            avg = (loss_info[0] / loss_info[1]).item()
            loss_info.zero_()
            if rank == 0:
                dt, t0 = time.time() - t0, time.time()
                print(f"step {completed}: train_loss {avg:.4f} ppl {math.exp(min(avg, 50)):.1f} lr {out['lr']:.3e} "
                      f"|g| {float(out['grad_norm']):.3f} {B * accum * world / dt:.1f} examples/s", flush=True)
            if completed % args.eval_every_n_steps == 0 or completed == max_steps:
                el, ea = evaluate()
                if rank == 0:
                    print(f"step {completed}: eval_loss {el:.4f} eval_teacher_acc {ea:.4f}", flush=True)
            if rank == 0 and ((ckpt_every and completed % ckpt_every == 0) or completed == max_steps):
                ck = os.path.join(args.output_dir, "final_checkpt" if completed == max_steps else f"step_{completed}")
                model.save_pretrained(ck)
                torch.save(tr.state_dict(), os.path.join(ck, "trainer_state.pt"))
            if completed >= max_steps:
                dist_mod.barrier()
                return


if __name__ == "__main__":
    main()
