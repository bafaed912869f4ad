"""Trainer of the SMILES VAE: the reference's interface (polygon/vae/vae_trainer.py:16-292) over the fused HIP step.

``VAETrainer(model, **kwargs).fit(train, val, n_epoch, batch_size, save_frequency)`` keeps the reference's defaults,
kwargs-override convention, CSV log, checkpoint cadence and quirks (``batch_size`` is ignored in favour of ``n_batch``;
a fresh Adam and fresh annealers per ``fit``).  The per-batch body -- forward, ``kl_w*kl + recon``, backward,
``clip_grad_norm_(50)``, ``Adam.step`` (vae_trainer.py:121-138) -- runs as five C-ABI calls with no autograd graph and,
under torch.distributed, an overlapped RCCL gradient all-reduce (polygon_amd/parallel.py).
"""
import ctypes
import logging
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from polygon_amd import _lib
from polygon_amd.parallel import GradComm, world_info
from polygon_amd.utils.moses_utils import CircularBuffer, set_torch_seed_to_all_gens
from polygon_amd.utils.utils import save_model
from polygon_amd.vae.vae_misc import CosineAnnealingLRWithRestart, KLAnnealer
from polygon_amd.vae.vae_model import TokenBatch, _Dp


class FusedClipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(defaults) + clip_grad_norm_ over the model's flat buffers in two kernels (pg_clip_adam).
    A regular torch Optimizer as far as schedulers are concerned (param_groups[0]['lr'])."""

    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, clip_grad=50.0):
        params = [p for p in model.vae.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.model = model
        self.clip_grad = clip_grad
        eng = model._engine
        flat = eng.ensure_flat()
        self.flat_grad = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scratch = torch.zeros(4096, device=flat.device)
        self.gnorm = torch.zeros((), device=flat.device)
        self.n_steps = 0
        # frozen parameters (freeze_embeddings): their gradient range is zeroed before the update
        self.frozen = []
        off = 0
        for p in model.vae.parameters():
            if not p.requires_grad:
                self.frozen.append((off, off + p.numel()))
            off += p.numel()

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model._engine
        flat = eng.ensure_flat()
        if flat.shape != self.exp_avg.shape or flat.device != self.exp_avg.device:
            raise RuntimeError("model parameters changed shape or device under the optimiser")
        for a, b in self.frozen:
            self.flat_grad[a:b].zero_()
        g = self.param_groups[0]
        self.n_steps += 1
        # pg_clip_adam_zero: the kernel leaves flat_grad zeroed (the reference's zero_grad(), vae_trainer.py:133, folded into
        # the update), and the engine is told so that the next fused step skips its own fill of the buffer
        _lib.check(_lib.lib().pg_clip_adam_zero(ctypes.c_long(flat.numel()), _lib.ptr(flat), _lib.ptr(self.flat_grad),
                                                _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.n_steps,
                                                ctypes.c_float(g['lr']), ctypes.c_float(g['betas'][0]),
                                                ctypes.c_float(g['betas'][1]), ctypes.c_float(g['eps']),
                                                ctypes.c_float(self.clip_grad if self.clip_grad else 0.0), _lib.ptr(self.scratch),
                                                _lib.ptr(self.gnorm), _lib.stream_ptr()), "pg_clip_adam_zero")
        eng.clean_grads = (self.flat_grad.data_ptr(), torch.cuda.current_stream(flat.device).cuda_stream)
        eng.invalidate()

    # ---- optimiser state for checkpoint / resume (no counterpart in the reference, which saves weights only:
    # polygon/utils/utils.py:90-110) ------------------------------------------------------------
    def train_state(self):
        return {"exp_avg": self.exp_avg.detach().cpu(), "exp_avg_sq": self.exp_avg_sq.detach().cpu(),
                "n_steps": self.n_steps, "lr": self.param_groups[0]["lr"], "n_params": int(self.exp_avg.numel())}

    def load_train_state(self, st):
        if int(st["n_params"]) != int(self.exp_avg.numel()):
            raise ValueError("optimiser state holds %d values, the model has %d" % (st["n_params"], self.exp_avg.numel()))
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.n_steps = int(st["n_steps"])
        self.param_groups[0]["lr"] = float(st["lr"])


class _Prefetcher:
    """Bounded background iteration of a DataLoader (see VAETrainer._prefetch)."""
    _END = object()

    def __init__(self, loader, depth):
        import queue
        import threading
        self.q = queue.Queue(maxsize=depth)
        self.stop = False
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def work():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)
                for b in loader:
                    if self.stop:
                        return
                    self.q.put(b)
                self.q.put(self._END)
            except BaseException as e:      # surfaces in the consumer
                self.q.put(e)
        self.t = threading.Thread(target=work, daemon=True, name="polygon-prefetch")
        self.t.start()

    def __iter__(self):
        return self

    def __next__(self):
        b = self.q.get()
        if b is self._END:
            raise StopIteration
        if isinstance(b, BaseException):
            raise b
        return b

    def close(self):
        self.stop = True
        try:
            while True:
                self.q.get_nowait()
        except Exception:
            pass

    def __del__(self):
        self.close()


class VAETrainer:
    def __init__(self, model, **kwargs):
        self.model = model
        # defaults (vae_trainer.py:21-39)
        self.log_file = None
        self.model_save = "model_{}.pt".format(int(time.time()))
        self.save_frequency = 1
        self.n_workers = 0
        self.n_batch = 512
        self.n_epoch = 10
        self.n_last = 1000
        self.clip_grad = 50
        self.lr_start = 0.0003
        self.kl_w_start = 0
        self.kl_start = 1
        self.kl_w_end = 1
        self.lr_n_period = 10
        self.lr_n_restarts = 6
        self.lr_n_mult = 1
        self.lr_end = 0.0003
        # additions (off by default = the reference's behaviour): write '<stem>_<epoch>.state.pt' (optimiser moments, step
        # count, epoch, scheduler position) next to every weight checkpoint; fit(resume_from=...) continues from one
        self.save_train_state = False
        # batches collated ahead of the GPU step on a background thread; 0 = in line, as the reference.  Measured on the
        # MI355X box (bench.py e2e leg, 128 batches of 1024): in-line collate costs 0.47 ms per batch and fit() runs at 98 %
        # of the resident-batch rate; the background thread competes with the launching thread for the GIL and fit() drops
        # to 90 % - so the default is in line.  Worth turning on only where the host is slow (collate >= the GPU step).
        self.prefetch = 0
        self.__dict__.update(kwargs)
        self.device = self.model.device

    def _n_epoch(self):
        if self.n_epoch:
            return self.n_epoch
        return sum(self.lr_n_period * (self.lr_n_mult ** i) for i in range(self.lr_n_restarts))

    def _save_model(self, output_path=None):
        save_model(self.model, self.model_save if output_path is None else output_path)

    def _shared_seed(self):
        """One seed all ranks agree on (rank 0's torch seed), drawn once per trainer: it keys the shuffle of the global
        batches and the eps noise, which must be the same on every rank for the shards to form one global batch."""
        if getattr(self, "_seed", None) is None:
            seed = torch.tensor([torch.initial_seed() % (2 ** 31)], dtype=torch.int64)
            rank, world = world_info()
            if world > 1:
                t = seed.cuda() if torch.distributed.get_backend() == "nccl" else seed
                GradComm().broadcast(t, 0)
                seed = t.cpu()
            self._seed = int(seed.item())
        return self._seed

    def get_dataloader(self, data, batch_size=None, collate_fn=None, shuffle=True):
        """vae_trainer.py:78-97.  ``batch_size`` is accepted and ignored exactly as the reference does (``n_batch`` is
        used).  Under data parallelism the loader draws GLOBAL batches of n_batch * world strings, identically shuffled
        on every rank (shared generator seed), and the collate keeps and tokenises only this rank's rows (sorted on
        lengths first: VAE.get_collate_fn(shard=...))."""
        rank, world = world_info()
        if collate_fn is None:
            collate_fn = self.model.get_collate_fn(shard=(rank, world) if world > 1 else None)
        gen = None
        if world > 1 and shuffle:
            gen = torch.Generator().manual_seed(self._shared_seed())
        return DataLoader(data, batch_size=self.n_batch * world, shuffle=shuffle, num_workers=self.n_workers,
                          collate_fn=collate_fn, generator=gen,
                          worker_init_fn=set_torch_seed_to_all_gens if self.n_workers > 0 else None)

    def _seed_gen(self):
        """Host generator seeded with the shared seed: every rank draws the same per-step dropout seeds."""
        if getattr(self, "_hgen", None) is None:
            self._hgen = torch.Generator().manual_seed(self._shared_seed() + 1)
        return self._hgen

    def _draw_eps(self, batch):
        """eps of vae_model.py:233 for this rank's rows.  Single process: torch.randn as the reference.  Data parallel:
        the GLOBAL batch's eps from a device generator that every rank seeds identically, then this rank's rows
        (rank, rank+world, ...), so that the shards form one global batch whatever the number of ranks."""
        rank, world = world_info()
        dev = self.model._engine.flat.device
        dz = self.model.d_z
        if world == 1 or batch.dp is None:
            return torch.randn(len(batch), dz, device=dev)
        if getattr(self, "_dgen", None) is None or self._dgen.device != dev:
            self._dgen = torch.Generator(device=dev).manual_seed(self._shared_seed() + 2)
        return torch.randn(batch.global_rows, dz, device=dev, generator=self._dgen)[rank::world].contiguous()

    def _prefetch(self, data_loader):
        """Iterates ``data_loader`` on a background thread, ``prefetch`` batches ahead (default 0 = in line): sampling,
        the length sort, tokenising and the pinned upload of batch i+1 overlap the GPU step of batch i (the launches are
        asynchronous and ctypes releases the GIL).  Order and contents are exactly the loader's."""
        depth = int(getattr(self, "prefetch", 0) or 0)
        if depth <= 0:
            return iter(data_loader)
        return _Prefetcher(data_loader, depth)

    def get_optim_params(self):
        return (p for p in self.model.vae.parameters() if p.requires_grad)

    # ---- one epoch (vae_trainer.py:102-189) -----------------------------------------------------------------
    def _train_epoch(self, epoch, data_loader, kl_weight, optimizer=None, label="Training"):
        t_epoch = time.process_time()
        model = self.model
        if optimizer is None:
            model.eval()
        else:
            model.train()
        n_batches = len(data_loader)
        kl_buf, rec_buf, loss_buf = CircularBuffer(n_batches), CircularBuffer(n_batches), CircularBuffer(n_batches)
        rank, world = world_info()
        fused = isinstance(optimizer, FusedClipAdam)
        comm = GradComm() if world > 1 else None
        eng = model._engine
        debug = logging.getLogger().isEnabledFor(logging.DEBUG)
        pending = []          # device scalars read back once per epoch unless per-batch debug logging is on
        lr = 0.
        # data parallel: no rank may raise in the middle of a step (its peers would wait in the all-reduce); a give-up is
        # shared through the gradient reduction (GradComm._share_failure) and raised by every rank at the epoch's end
        checks = _lib.lib().pg_entry_checks(0) if world > 1 else None
        try:
            return self._epoch_body(epoch, data_loader, kl_weight, optimizer, label, t_epoch, n_batches, kl_buf, rec_buf, loss_buf,
                                    rank, world, fused, comm, eng, debug, pending, lr)
        finally:
            if checks is not None:
                _lib.lib().pg_entry_checks(checks)

    def _epoch_body(self, epoch, data_loader, kl_weight, optimizer, label, t_epoch, n_batches, kl_buf, rec_buf, loss_buf,
                    rank, world, fused, comm, eng, debug, pending, lr):
        model = self.model
        for i, batch in enumerate(self._prefetch(data_loader)):
            t_batch = time.process_time()
            if world > 1:
                if not isinstance(batch, TokenBatch) or batch.dp is None:
                    raise RuntimeError("data-parallel training needs the sharding collate (VAETrainer.get_dataloader)")
                if batch.global_rows < world:
                    continue   # a tail with fewer strings than ranks would leave some ranks without rows: skipped on EVERY rank
            if fused or world > 1:
                eng.ensure_flat()      # (the shadow weights are rebuilt inside train_step / eval_losses, under deferral)
                b = batch if isinstance(batch, TokenBatch) else None
                if b is None:
                    from polygon_amd.vae.vae_model import _as_batch
                    b = _as_batch(batch, model.pad, eng.flat.device)
                eps = self._draw_eps(b)
                if optimizer is not None:
                    p_drop = float(model.d_dropout) if model.d_n_layers > 1 else 0.0
                    seed = int(torch.randint(0, 2 ** 62, (1,), generator=self._seed_gen()).item()) if p_drop > 0 else 0
                    q_drop = float(model.q_dropout) if model.q_n_layers > 1 else 0.0     # nn.GRU's own rule (vae_model.py:69)
                    q_seed = int(torch.randint(0, 2 ** 62, (1,), generator=self._seed_gen()).item()) if q_drop > 0 else 0
                    kl_loss, recon_loss = eng.train_step(b, kl_weight, optimizer.flat_grad, eps=eps, p_drop=p_drop,
                                                         seed=seed, comm=comm, q_drop=q_drop, q_seed=q_seed)
                    optimizer.step()
                else:     # validation (vae_trainer.py:246-252: eval mode, fresh eps), forward only
                    kl_loss, recon_loss = eng.eval_losses(b, eps=eps)
                loss = kl_weight * kl_loss + recon_loss
            else:
                kl_loss, recon_loss = model(batch)
                loss = kl_weight * kl_loss + recon_loss
                if optimizer is not None:
                    optimizer.zero_grad()
                    loss.backward()
                    torch.nn.utils.clip_grad_norm_(self.get_optim_params(), self.clip_grad)
                    optimizer.step()
            lr = optimizer.param_groups[0]['lr'] if optimizer is not None else 0.
            vals = torch.stack([kl_loss.detach().reshape(()), recon_loss.detach().reshape(()), loss.detach().reshape(())])
            if debug:
                if comm is not None:
                    comm.sum_scalars(vals)      # per-rank shares of the global-batch losses -> the global values
                k, r, l = vals.tolist()
                kl_buf.add(k); rec_buf.add(r); loss_buf.add(l)
                logging.debug(' '.join([label, f'Batch {str(i).zfill(len(str(n_batches)))}/{n_batches}',
                                        f'Time={time.process_time() - t_batch:.2f}', f'loss={l:.5f}', f'(kl={k:.5f}',
                                        f'recon={r:.5f})', f'klw={kl_weight:.5f}', f'lr={lr:.5f}']))
            else:
                pending.append(vals)
        if pending:
            allv = torch.stack(pending)
            if comm is not None:
                comm.sum_scalars(allv)          # one all-reduce per epoch
            for k, r, l in allv.tolist():
                kl_buf.add(k); rec_buf.add(r); loss_buf.add(l)
        _lib.check(_lib.lib().pg_check_failure(_lib.stream_ptr()), "training epoch")   # after the epoch's sync, before anything is saved
        kl_v, rec_v, loss_v = kl_buf.mean(), rec_buf.mean(), loss_buf.mean()
        logging.info(' '.join([label, f'Time={time.process_time() - t_epoch:.2f}', f'loss={loss_v:.5f}', f'(kl={kl_v:.5f}',
                               f'recon={rec_v:.5f})', f'klw={kl_weight:.5f}', f'lr={lr:.5f}']))
        return {'epoch': epoch, 'kl_weight': kl_weight, 'lr': lr, 'kl_loss': kl_v, 'recon_loss': rec_v, 'loss': loss_v,
                'mode': 'Eval' if optimizer is None else 'Train'}

    def _log_epoch(self, postfix, header=False):
        columns = ['epoch', 'kl_weight', 'lr', 'kl_loss', 'recon_loss', 'loss', 'mode']
        if self.log_file is None:
            return
        with open(self.log_file, "a+") as handle:
            handle.write(",".join(columns if header else [str(postfix[c]) for c in columns]) + "\n")

    def _make_optimizer(self):
        """Adam(lr_start) as vae_trainer.py:217-218; fused when the model lives on the GPU (always, in practice)."""
        return FusedClipAdam(self.model, lr=self.lr_start, clip_grad=self.clip_grad)

    def _train(self, train_loader, val_loader=None, n_epoch=None, save_frequency=None, resume_from=None):
        if n_epoch is None:
            n_epoch = self._n_epoch()
        if save_frequency is None:
            save_frequency = self.save_frequency
        optimizer = self._make_optimizer()
        kl_annealer = KLAnnealer(n_epoch, kl_start=self.kl_start, kl_w_end=self.kl_w_end, kl_w_start=self.kl_w_start)
        lr_annealer = CosineAnnealingLRWithRestart(optimizer, lr_n_period=self.lr_n_period, lr_n_restarts=self.lr_n_restarts,
                                                   lr_n_mult=self.lr_n_mult, lr_end=self.lr_end)
        self.model.zero_grad()
        rank, _ = world_info()
        width = len(str(n_epoch))
        first_epoch = 0
        if resume_from is not None:   # the weights themselves come from load_model(); this restores the rest
            st = torch.load(resume_from, map_location="cpu", weights_only=True)
            optimizer.load_train_state(st["optimizer"])
            for _ in range(int(st["epoch"]) + 1):
                lr_annealer.step()    # the schedule is a pure function of the epoch count (vae_misc.py)
            optimizer.param_groups[0]["lr"] = float(st["lr_next"])
            first_epoch = int(st["epoch"]) + 1
        for epoch in range(first_epoch, n_epoch):
            kl_weight = kl_annealer(epoch)
            postfix = self._train_epoch(epoch, train_loader, kl_weight, optimizer,
                                        label='Training   (epoch #{})'.format(str(epoch).zfill(width)))
            if rank == 0:
                self._log_epoch(postfix)
            if val_loader is not None:
                postfix = self._train_epoch(epoch, val_loader, kl_weight,
                                            label='Validation (epoch #{})'.format(str(epoch).zfill(width)))
                if rank == 0:
                    self._log_epoch(postfix)
            if (self.model_save is not None) and (epoch % save_frequency == 0) and rank == 0:
                save_model(self.model, output_path='{path}_{epoch:03d}.pt'.format(
                    path=os.path.splitext(self.model_save)[0], epoch=epoch))
            lr_annealer.step()
            if self.save_train_state and (self.model_save is not None) and (epoch % save_frequency == 0) and rank == 0:
                torch.save({"epoch": epoch, "optimizer": optimizer.train_state(),
                            "lr_next": optimizer.param_groups[0]["lr"]},
                           '{path}_{epoch:03d}.state.pt'.format(path=os.path.splitext(self.model_save)[0], epoch=epoch))

    def fit(self, train_data, val_data=None, n_epoch=None, batch_size=None, save_frequency=None, resume_from=None):
        rank, _ = world_info()
        if self.log_file is not None and rank == 0:
            lines = ["Training Parameters:", "\n".join("{}: {}".format(k, v) for k, v in self.__dict__.items()),
                     "----------------------"]
            with open(self.log_file, "w") as handle:
                handle.write("\n".join(lines) + "\n")
        _, world = world_info()
        if world > 1:     # every rank starts from rank 0's weights (the reference has one process; nothing else syncs them)
            flat = self.model._engine.ensure_flat()
            GradComm().broadcast(flat, 0)
            self.model._engine.invalidate()
        train_loader = self.get_dataloader(train_data, batch_size=batch_size, shuffle=True)
        val_loader = None if val_data is None else self.get_dataloader(val_data, batch_size=batch_size, shuffle=False)
        self._train(train_loader, val_loader, n_epoch=n_epoch, save_frequency=save_frequency, resume_from=resume_from)
        if self.model_save is not None and rank == 0:
            save_model(self.model, output_path=self.model_save)
