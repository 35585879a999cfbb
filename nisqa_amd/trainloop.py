"""Host side of ``nisqaModel.train()`` (reference nisqa/NISQA_model.py:83-570, ``_train_mos`` / ``_train_dim``): epochs,
shuffled mini-batches, the bias-aware loss bookkeeping, per-epoch evaluation on the training and validation sets,
ReduceLROnPlateau, early stopping, results CSV and checkpoints -- around the per-batch step, which runs as HIP kernels
(``nisqa_amd.train.HipTrainer``: forward in train mode, backward, Adam).  WAV files reach the GPU through the same
native ingest as prediction (``nisqa_amd.ingest``); the validation pass is the inference engine on the current weights.

One loop serves both model types: ``targets`` is ['mos'] (well: csv_mos_train) for NISQA and
['mos', 'noi', 'dis', 'col', 'loud'] for NISQA_DIM, with the reference's key suffixes ('', '_noi', ...) in the result
dicts, its printed lines and its checkpoint dictionary (so ``run_predict.py`` loads what this writes, here and in the
reference).
"""
import os
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch
import yaml

from . import NISQA_lib as NL
from . import dist as _dist
from . import ingest as _ingest
from .evaluation import eval_results
from .train import HipTrainer


# ---- mirrors of the reference's small training helpers ------------------------------------------------------------
class biasLoss(object):
    """Per-database first-order bias of the predictions (reference NISQA_lib.py:1855-1950): once the epoch's Pearson
    correlation exceeds ``min_r`` every database gets the line mapping its predictions onto its labels, and the loss
    is taken after that mapping (``b`` rows of the samples in the batch go to the loss kernel)."""

    def __init__(self, db, anchor_db=None, mapping='first_order', min_r=0.7, loss_weight=0.0, do_print=True):
        self.db, self.mapping, self.min_r, self.anchor_db = db, mapping, min_r, anchor_db
        self.loss_weight, self.do_print = loss_weight, do_print
        self.b = np.zeros((len(db), 4))
        self.b[:, 1] = 1
        self.do_update = False
        self.apply_bias_loss = not (min_r is None or mapping is None)
        if self.apply_bias_loss and loss_weight:
            raise NotImplementedError('biasLoss loss_weight != 0 is not built (0.0 in the reference\'s training code)')

    def rows(self, idx):
        """[len(idx), 4] cubic coefficients for the loss kernel, or None while no bias is applied."""
        return self.b[np.asarray(idx)].astype(np.float32) if self.apply_bias_loss else None

    def update_bias(self, y, y_hat):
        if not self.apply_bias_loss:
            return
        from scipy.stats import pearsonr
        y, y_hat = np.asarray(y).reshape(-1), np.asarray(y_hat).reshape(-1)
        if not self.do_update:
            ok = ~np.isnan(y)
            r = pearsonr(y[ok], y_hat[ok])[0]
            if self.do_print:
                print('--> bias update: min_r {:0.2f}, r_p {:0.2f}'.format(r, self.min_r))
            if r > self.min_r:
                self.do_update = True
        if self.do_update:
            if self.do_print:
                print('--> bias updated')
            if self.mapping != 'first_order':
                raise NotImplementedError
            for db_name in self.db.unique():
                sel = (self.db == db_name).to_numpy().nonzero()[0]
                if not np.isnan(y[sel]).any() and db_name != self.anchor_db:
                    a = np.vstack([np.ones(len(sel)), y_hat[sel]]).T
                    self.b[sel, :2] = np.linalg.lstsq(a, y[sel], rcond=None)[0]


class earlyStopper(object):
    """Stop when neither r_p nor the mapped RMSE of any head improved for ``patience`` epochs (NISQA_lib.py:1941-2041);
    ``best`` marks an epoch with a new best MOS RMSE (checkpoint policy 'best_only')."""

    def __init__(self, patience, suffixes=('',)):
        self.suffixes, self.patience = suffixes, patience
        self.best_r = {s: -1e10 for s in suffixes}
        self.best_e = {s: 1e10 for s in suffixes}
        self.cnt, self.best = -1, False

    best_r_p = property(lambda self: self.best_r[''])
    best_rmse = property(lambda self: self.best_e[''])

    def step(self, r):
        self.best = False
        for s in self.suffixes:
            if r['r_p_mean_file' + s] > self.best_r[s]:
                self.best_r[s] = r['r_p_mean_file' + s]
                self.cnt = -1
            if r['rmse_map_mean_file' + s] < self.best_e[s]:
                self.best_e[s] = r['rmse_map_mean_file' + s]
                self.cnt = -1
                if s == '':
                    self.best = True
        self.cnt += 1
        return self.cnt >= self.patience


class ReduceLROnPlateau(object):
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.1, threshold=0.003 relative, cooldown 0,
    min_lr 0, eps 1e-8) as configured at NISQA_model.py:97-102, acting on ``trainer.lr``."""

    def __init__(self, trainer, patience, threshold=0.003, factor=0.1):
        self.tr, self.patience, self.threshold, self.factor = trainer, patience, threshold, factor
        self.best, self.bad = float('inf'), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = self.tr.lr * self.factor
            if self.tr.lr - new > 1e-8:
                self.tr.lr = new
                print('Reducing learning rate to {:.4e}.'.format(new))
            self.bad = 0


# ---- the loop -----------------------------------------------------------------------------------------------------
_DIM = ['mos', 'noi', 'dis', 'col', 'loud']


def _targets(nm):
    return list(_DIM) if nm.args['dim'] else [nm.args['csv_mos_train']]


def _evaluate(nm, ds, targets, preds, target_names, label):
    """eval_results per head with the reference's prints -> (list of per-db frames, list of overall dicts)"""
    verbose = nm.args['tr_verbose'] > 0
    dbs, rs = [], []
    for t, p, name in zip(targets, preds, target_names):
        if verbose and len(targets) > 1:
            print('--> %s:' % name.upper())
        db, r = eval_results(ds.df, dcon=ds.df_con, target_mos=t, target_ci=t + '_ci', pred=p, mapping='first_order',
                             do_print=verbose)
        dbs.append(db)
        rs.append(r)
    return dbs, rs


def train(nm):
    """Body of ``nisqaModel.train()``; ``nm`` is the nisqaModel (args, model, ds_train, ds_val, dev)."""
    a = nm.args
    dim = bool(a['dim'])
    targets = _targets(nm)
    val_targets = list(_DIM) if dim else [a['csv_mos_val']]
    names = _DIM if dim else ['mos']
    preds = [n + '_pred' for n in names]
    sfx = [''] + ['_' + n for n in names[1:]]
    nm.runname = nm._makeRunnameAndWriteYAML()
    tr = HipTrainer(a, nm.model.state_dict(), nm.dev, lr=a['tr_lr'])
    scheduler = ReduceLROnPlateau(tr, a['tr_lr_patience'])
    stopper = earlyStopper(a['tr_early_stop'], tuple(sfx))
    losses = [biasLoss(nm.ds_train.df.db, anchor_db=a['tr_bias_anchor_db'], mapping=a['tr_bias_mapping'],
                       min_r=a['tr_bias_min_r'], do_print=(a['tr_verbose'] > 0)) for _ in names]
    n_train, bs = len(nm.ds_train), int(a['tr_bs'])
    y_train = np.stack([nm.ds_train.df[t].to_numpy(dtype=np.float64) for t in targets], 1)
    rng = np.random.default_rng(int(torch.initial_seed()) & 0xffffffff)      # the same permutation on every rank
    rank, world = _dist.world()

    print('--> start training')
    for epoch in range(a['tr_epochs']):
        tic = time.time()
        order = rng.permutation(n_train)                                 # DataLoader(shuffle=True, drop_last=False)
        batches = [order[s:s + bs].tolist() for s in range(0, n_train, bs)]
        if world > 1:                                                    # data parallel: every rank takes its share of each batch
            if len(batches) > 1 and len(batches[-1]) < world:
                batches[-2:] = [batches[-2] + batches[-1]]
            if len(batches[-1]) < world:
                raise ValueError('fewer training files ({}) than ranks ({})'.format(n_train, world))
            batches = [b[rank::world] for b in batches]
        y_hat_train = np.zeros((n_train, len(names)))
        loss_sum = 0.0
        ing = _ingest.Ingest(nm.ds_train, batches, pin=True, num_workers=a['tr_num_workers'])
        pending = None                                                   # (idx, device y_hat, device loss) of the last step
        try:
            for staged in ing:
                if len(staged.groups) != 1:
                    ing.ring.release_after(staged.slot, None)
                    raise NotImplementedError('a training batch mixes sample rates {}: BatchNorm statistics span the batch, '
                                              'so it cannot be split'.format([g.sr for g in staged.groups]))
                g = staged.groups[0]
                plan = tr.eng.audio_plan(g.lengths, g.sr, names=g.names)     # (at ms_sr when the run sets it: lb.load resamples)
                raw = ing.ring.buf[staged.slot]
                host = raw[g.offset:g.offset + g.nbytes].view(torch.int16 if g.is_i16 else torch.float32)
                pcm = host.to(tr.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                ing.ring.release_after(staged.slot, ev)
                pcm = tr.eng.resample(pcm, g.lengths, g.sr)              # a no-op unless ms_sr is set and differs from the files' rate
                if pcm.dtype == torch.int16:
                    pcm = tr.eng.pcm16_to_f32(pcm)
                ids = np.asarray(g.ids)
                bias = None
                if losses[0].apply_bias_loss:
                    if len(names) > 1:
                        raise NotImplementedError('bias loss for NISQA_DIM needs one mapping per head (not built)')
                    bias = losses[0].rows(ids)
                if pending is not None:                                  # fetch the previous step's numbers while this one runs
                    y_hat_train[pending[0]] = pending[1].cpu().numpy()
                    loss_sum += float(pending[2])
                loss = tr.step_pcm(pcm, plan, tr.eng.rate(g.sr), y_train[ids].astype(np.float32), bias=bias)
                pending = (ids, tr.last['y_hat'], loss)
            if pending is not None:
                y_hat_train[pending[0]] = pending[1].cpu().numpy()
                loss_sum += float(pending[2])
        finally:
            ing.close()
        loss = loss_sum / max(1, len(batches))
        if world > 1:                                                    # ranks filled disjoint rows of the prediction table
            y_hat_train = _dist.all_reduce_sum_(torch.from_numpy(y_hat_train)).numpy()
        for h, bl in enumerate(losses):
            bl.update_bias(y_train[:, h].reshape(-1, 1), y_hat_train[:, h].reshape(-1, 1))

        # ---- evaluation on the training predictions and on the validation set (inference engine, current weights)
        if a['tr_verbose'] > 0:
            print('\n<---- Training ---->')
        for h, p in enumerate(preds):
            nm.ds_train.df[p] = y_hat_train[:, h].reshape(-1, 1)
        _, r_tr = _evaluate(nm, nm.ds_train, targets, preds, names, 'train')
        nm.model.load_state_dict(tr.state_dict(), strict=True)
        nm.model.bind_args(a)                                            # drops the cached engine: next predict repacks
        if a['tr_verbose'] > 0:
            print('<---- Validation ---->')
        (NL.predict_dim if dim else NL.predict_mos)(nm.model, nm.ds_val, a['tr_bs_val'], nm.dev,
                                                    num_workers=a['tr_num_workers'])
        db_val, r_val = _evaluate(nm, nm.ds_val, val_targets, preds, names, 'val')
        r = {}
        for s, rt in zip(sfx, r_tr):
            r['train_r_p_mean_file' + s] = rt['r_p_mean_file']
            r['train_rmse_map_mean_file' + s] = rt['rmse_map_mean_file']
        for s, rv in zip(sfx, r_val):
            r.update({k + s: v for k, v in rv.items()})
        db_results = {'db_results_val_' + n: d for n, d in zip(names, db_val)} if dim else db_val[0]

        scheduler.step(loss)
        stop = stopper.step(r)
        ep_runtime = time.time() - tic
        mid = ''
        if dim:
            mid = 'r_dim_mos_mean {:0.2f}, '.format(sum(r['r_p_mean_file' + s] for s in sfx) / 5)
        print('ep {} sec {:0.0f} es {} lr {:0.0e} loss {:0.4f} // r_p_tr {:0.2f} rmse_map_tr {:0.2f} // {}r_p {:0.2f} '
              'rmse_map {:0.2f} // best_r_p {:0.2f} best_rmse_map {:0.2f},'
              .format(epoch + 1, ep_runtime, stopper.cnt, tr.lr, loss, r['train_r_p_mean_file'],
                      r['train_rmse_map_mean_file'], mid, r['r_p_mean_file'], r['rmse_map_mean_file'], stopper.best_r_p,
                      stopper.best_rmse))
        if rank == 0:
            _save_results(nm, tr, epoch, loss, ep_runtime, r, db_results, stopper.best)
        if stop:
            print('--> Early stopping. best_r_p {:0.2f} best_rmse {:0.2f}'.format(stopper.best_r_p, stopper.best_rmse))
            return
    print('--> Training done. best_r_p {:0.2f} best_rmse_map {:0.2f}'.format(stopper.best_r_p, stopper.best_rmse))


def _save_results(nm, tr, epoch, loss, ep_runtime, r, db_results, best):
    """Results CSV + checkpoint dictionary with the reference's keys (NISQA_model.py:1053-1111)."""
    a = nm.args
    if a['tr_checkpoint'] not in ('every_epoch', 'best_only', None):
        raise ValueError('selected tr_checkpoint option not available')
    filename = nm.runname + ('.tar' if a['tr_checkpoint'] == 'best_only' else '__ep_{:03d}.tar'.format(epoch + 1))
    out_dir = os.path.join(a['output_dir'], nm.runname)
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    results = {'runname': nm.runname, 'epoch': '{:05d}'.format(epoch + 1), 'filename': filename, 'loss': loss,
               'ep_runtime': '{:0.2f}'.format(ep_runtime)}
    results.update(nm.runinfos)
    results.update(r)
    results.update(a)
    results = {k: str(v) for k, v in results.items()}
    if epoch == 0:
        nm.results_hist = pd.DataFrame(results, index=[0])
    else:
        nm.results_hist.loc[epoch] = results
    nm.results_hist.to_csv(os.path.join(out_dir, nm.runname + '__results.csv'), index=False)
    if a['tr_checkpoint'] == 'every_epoch' or (a['tr_checkpoint'] == 'best_only' and best):
        torch.save({'runname': nm.runname, 'epoch': epoch + 1, 'model_args': _plain(nm.model_args), 'args': _plain(a),
                    'model_state_dict': tr.state_dict(),
                    'optimizer_state_dict': {'step': tr.t, 'lr': tr.lr, 'exp_avg': tr.m.cpu(), 'exp_avg_sq': tr.v.cpu(),
                                             'layout': 'nisqa_amd flat buffer (HipTrainer.keys / kshape order)'},
                    'db_results': _plain(db_results), 'results': _plain(results), 'model_name': nm.model.name},
                   os.path.join(out_dir, filename))


def _plain(x):
    """Containers with numpy scalars / arrays -> plain Python numbers and lists, so that a checkpoint written here loads
    through torch's restricted unpickler (NISQA_model._load_checkpoint) -- numpy's pickle constructors are not on its list."""
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, pd.DataFrame):
        return _plain(x.to_dict(orient='list'))
    return x


def make_runname_and_write_yaml(nm):
    """NISQA_model.py:718-730"""
    runname = nm.args['name'] + '_' + nm.args['now'].strftime('%y%m%d_%H%M%S%f')
    print('runname: ' + runname)
    out_dir = os.path.join(nm.args['output_dir'], runname)
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    with open(os.path.join(out_dir, runname + '.yaml'), 'w') as f:
        yaml.dump(nm.args, f, default_flow_style=None, sort_keys=False)
    return runname
