"""Shared body of main_1d.py / main_2d.py: compose config, build data, model,
optimiser and scheduler, train, test, checkpoint -- the sequence of the
reference's entry points (main_1d.py:34-309, main_2d.py:38-324) minus wandb and
plotting.  Datasets: synthetic Markov pairs (default), or a file through the
reference's `dataset_params` convention (conf/dataset/ns/ns_file.yaml ->
dataloaders/ns_naive_markov.py)."""
from __future__ import annotations

import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.optim as optim

from rpde.config import compose, instantiate


def _init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RPDE_FORCE_DIST=1: join the process group even as a single rank, so that a 1-GPU box exercises the RCCL path
    # (librccl load, communicator init, the gradient all-reduce on the launch stream): tests/test_gpu_rccl.py
    force = os.environ.get("RPDE_FORCE_DIST") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    return world, rank, local


def run(dims: int, argv=None):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = compose(os.path.join(here, "conf"), "config", list(sys.argv[1:] if argv is None else argv))
    if int(args.dataset.dims) != dims:
        raise SystemExit(f"main_{dims}d.py needs a {dims}-D dataset config, got dims={args.dataset.dims}")
    world, rank, local = _init_distributed()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    from rpde.launch import limit_host_threads
    limit_host_threads(world)                  # host ops (batch stacking, pinned copies) on the cores this rank may use

    from train.mres_training import ResolutionGroupedDataLoader
    from train.training import evaluate, train
    from utils.synthetic import markov_pairs

    seed = int(args.training.get("seed", 0))
    bs = int(args.training.batch_size)
    x_normalizer = y_normalizer = min_data = max_data = min_model = max_model = rollout_set = None
    normalization_type = "simple"
    if args.dataset.get("dataset_params"):
        # reference main_2d.py:72-82: (train, val, test, *stats); main_1d.py:69-80: (train, val, test, rollout, *stats).
        # stats are (x_normalizer, y_normalizer) or, for Burgers' "minmax", (min_data, max_data, min_model, max_model)
        data_ = instantiate(args.dataset.dataset_params)
        train_set, val_set, test_set = data_[:3]
        if dims == 1:
            rollout_set = data_[3]
        stats = data_[4 if dims == 1 else 3:]
        if len(stats) == 4:
            normalization_type = "minmax"
            min_data, max_data, min_model, max_model = stats
        elif len(stats) == 2:
            x_normalizer, y_normalizer = stats
    else:
        train_set = markov_pairs(args.dataset.resolutions, dims, seed)
        top = max(int(r) for r in dict(args.dataset.resolutions))
        val_set = markov_pairs({top: int(args.dataset.n_val)}, dims, seed + 50000)
        test_set = markov_pairs({top: int(args.dataset.n_test)}, dims, seed + 60000)
    # training: equal rank slices of whole global batches; validation / test: every sample once (ragged rank shares)
    mk = lambda ds, shuffle: ResolutionGroupedDataLoader(ds, bs, shuffle=shuffle, seed=seed, rank=rank,  # noqa: E731
                                                         world_size=world, verbose=rank == 0, drop_last=shuffle)
    train_loader, val_loader, test_loader = mk(train_set, True), mk(val_set, False), mk(test_set, False)

    torch.manual_seed(seed)                                   # same initial weights on every rank
    model = instantiate(args.model).to(device)
    torch.manual_seed(seed + 1 + rank)                        # dropout seeds are drawn from this stream: one per rank
    ckpt = args.dataset.get("saved_checkpoint_path")
    if ckpt:
        state = torch.load(ckpt, map_location=device, weights_only=True)
        model.load_state_dict(state["model_state_dict"])

    # plans for every grid the run will meet -- the training / validation / test groups and the post-training sweep
    # [32 .. max] -- before the first step (no hipMalloc / stream sync inside the training loop)
    from rpde.ops import warm_plans
    from utils.resize_utils import get_lower_resolutions
    seen = {r for ld in (train_loader, val_loader, test_loader) for r in ld.resolution_groups}
    if seen:
        seen |= set(get_lower_resolutions(max(seen), min(32, max(seen))))
        warm_plans(model, seen, dims, in_channels=int(train_set[0][0].shape[0]), device=device)

    lr = float(args.training.learning_rate)
    # torch.optim.AdamW's update rule and state_dict format, one kernel per step (rpde/optim.py)
    from rpde.optim import FlatAdamW
    # training.graph=true (or RPDE_TRAIN_GRAPH=1): train() replays each batch shape's step as one hipGraph -- the optimizer's
    # step state (count, learning rate, weight decay) then lives on the device
    use_graph = bool(args.training.get("graph", False)) or os.environ.get("RPDE_TRAIN_GRAPH") == "1"
    if dims == 2:     # reference main_2d.py:173-174
        optimizer = FlatAdamW(model.parameters(), lr=lr, capturable=use_graph)
        scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=30, gamma=0.5)
    else:             # reference main_1d.py:144-145
        optimizer = FlatAdamW(model.parameters(), lr=lr, weight_decay=1e-4, capturable=use_graph)
        scheduler = optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=100, eta_min=1e-5)

    n_params = sum(p.numel() for p in model.parameters())
    if rank == 0:
        print(json.dumps({"model": args.model["_target_"], "params": n_params, "world": world,
                          "train_batches": len(train_loader), "choices": args["_choices_"]}), flush=True)
    from rpde.launch import freeze_setup_garbage
    freeze_setup_garbage()                     # model, optimizer, datasets are long-lived: keep full collections off them
    t0 = time.time()
    loss_hist, val_hist = train(model, train_loader, val_loader, optimizer, scheduler, y_normalizer=y_normalizer,
                                use_normalizer=bool(args.training.use_normalizer), epochs=int(args.training.epochs),
                                device=device, graph=use_graph)
    torch.cuda.synchronize()
    test_l2 = evaluate(model, test_loader, normalization_type=normalization_type, min_data=min_data, max_data=max_data,
                       min_model=min_model, max_model=max_model, y_normalizer=y_normalizer, device=device)
    if rank == 0:
        print(json.dumps({"train_seconds": round(time.time() - t0, 3), "final_train_loss": loss_hist[-1],
                          "final_val_loss": val_hist[-1], "test_rel_l2": test_l2}), flush=True)

    # ---- the reference's post-training sequence: every resolution [32, .., max] (main_2d.py:287, main_1d.py:250),
    # ---- and in 1-D the autoregressive rollout (main_1d.py:272; utils/autoregressive_step.py:284-309)
    from utils.resize_utils import evaluate_all_resolutions, to_resolution
    dp = args.dataset.get("dataset_params") or {}
    raw_eval = None
    if dp.get("eval_dataset_target") and args.dataset.get("train_mres"):
        # multi-resolution training evaluates on ONE single-resolution file through another loader (reference
        # utils/naive_utils.py:322-350): un-normalised pairs, encoded here with the training statistics
        keep = ("reduced_batch", "reduced_resolution_t", "use_low_pass_filter", "lowpass_cutoff_ratio", "num_samples_max")
        node = {"_target_": dp["eval_dataset_target"], "filename": dp["eval_filename"],
                "saved_folder": dp.get("eval_saved_folder", dp.get("saved_folder")), "data_normalizer": False,
                **{k: dp[k] for k in keep if k in dp}}
        raw_eval = instantiate(node)[2]
    pairs = [(raw_eval or test_set)[i] for i in range(len(raw_eval or test_set))]
    top_res = max(int(x.shape[-1]) for x, _ in pairs)
    pairs = [(torch.as_tensor(x), torch.as_tensor(y)) for x, y in pairs if int(x.shape[-1]) == top_res]   # a mixed test
    tx, ty = torch.stack([x for x, _ in pairs]), torch.stack([y for _, y in pairs])        # split: highest resolution only
    if raw_eval is not None and x_normalizer is not None:
        tx, ty = x_normalizer.encode(tx), y_normalizer.encode(ty)
    how = str(args.dataset.get("evaluation_type", "naive_downsample"))
    if y_normalizer is not None:
        dec = lambda t: y_normalizer.decode(t, device=device)                                    # noqa: E731
    elif min_model is not None:
        dec = lambda t: t * (max_model - min_model) + min_model                                  # noqa: E731
    else:
        dec = None
    ty_phys = dec(ty.to(device)).cpu() if dec else ty
    resolution_results = evaluate_all_resolutions(model, tx, ty_phys, max_resolution=top_res, min_resolution=min(32, top_res),
                                                  how=how, batch_size=bs, y_decode=dec, device=device)
    rollout_results = None
    if dims == 1:
        from utils.autoregressive_step import perform_rollout_1d, rollout_loss
        steps = int(args.dataset.get("rollout_steps", 4))
        if rollout_set is not None and len(rollout_set):
            # whole test trajectories of the file [T, X], physical units: encode the start, roll in normalised space
            # (decode with y-statistics, re-encode with x-statistics between steps), compare in physical units
            steps = min(steps, int(rollout_set[0].shape[0]) - 1)
            # (a multi-resolution loader hands over test trajectories of every file resolution: like the mixed test
            #  split above, the rollout is evaluated on the highest one)
            full = [rollout_set[i] for i in range(len(rollout_set)) if int(rollout_set[i].shape[-1]) == top_res]
            mine = full[rank::world]                                                             # this rank's trajectories
            phys = (torch.stack(mine)[:, :steps + 1] if mine else torch.zeros(0, steps + 1, top_res)).to(device)
            if min_data is not None:
                enc = lambda t: (t - min_data) / (max_data - min_data)                           # noqa: E731
                from types import SimpleNamespace
                xn = SimpleNamespace(encode=enc)
                yn = SimpleNamespace(decode=lambda t, device=None: t * (max_model - min_model) + min_model)
            else:
                xn, yn = x_normalizer, y_normalizer
                enc = xn.encode if xn is not None else (lambda t: t)                             # noqa: E731
            traj, decode_pred = phys, (lambda t: yn.decode(t, device=device)) if yn is not None else (lambda t: t)
        else:
            from utils.synthetic import advance
            frames = [tx[rank::world, 0]]
            for _ in range(steps):
                frames.append(advance(frames[-1], 1))
            traj = torch.stack(frames, dim=1).to(device)                       # [n, steps+1, res]
            xn = yn = None
            enc = decode_pred = lambda t: t                                      # noqa: E731
        acc = torch.zeros(len(resolution_results), 2, dtype=torch.float64, device=device)
        for k, res in enumerate(resolution_results):
            tr = to_resolution(traj, res, "naive_downsample")
            if tr.shape[0]:
                pred = perform_rollout_1d(model.eval(), enc(tr[:, 0]), steps, device=device, x_normalizer=xn, y_normalizer=yn)
                acc[k, 0] += rollout_loss(decode_pred(pred), tr) * tr.shape[0]
                acc[k, 1] += tr.shape[0]
        if world > 1:
            dist.all_reduce(acc)
        rollout_results = {res: (float(acc[k, 0] / acc[k, 1]) if float(acc[k, 1]) > 0 else float("nan"))
                           for k, res in enumerate(resolution_results)}
    if rank == 0:
        print(json.dumps({"evaluation_type": how, "resolution_rel_l2": {str(k): v for k, v in resolution_results.items()}}),
              flush=True)
        if rollout_results is not None:
            print(json.dumps({"rollout_rel_l2": {str(k): v for k, v in rollout_results.items()}}), flush=True)
    run.last = {"test_rel_l2": test_l2, "resolution_rel_l2": resolution_results, "rollout_rel_l2": rollout_results}
    if rank == 0:
        os.makedirs(args.checkpoint_dir, exist_ok=True)
        path = os.path.join(args.checkpoint_dir, f"{args.project_name}_{dims}d.pt")
        torch.save({"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                    "loss_history": loss_hist, "val_loss_history": val_hist, "l2_loss": test_l2,
                    "resolution_rel_l2": resolution_results}, path)
        print(json.dumps({"checkpoint": path}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return test_l2
