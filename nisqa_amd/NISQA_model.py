"""Drop-in for the reference's ``nisqa/NISQA_model.py`` on the predict path: same class name,
constructor argument dict, attributes (``.args .dev .model .ds_val``), printed lines, result frame
and ``NISQA_results.csv`` -- with the hot path behind it running as HIP kernels on an MI355X.

What ``run_predict.py`` and ``run_evaluate.py`` reach is implemented (reference NISQA_model.py:26-81, 572-716,
732-847, 928-1051): the three predict modes and ``evaluate()`` on their predictions (host-side P.1401 statistics,
nisqa_amd/evaluation.py).  ``train()`` runs the reference's epoch loop (nisqa_amd/trainloop.py) around the HIP training step.
"""
import datetime
import os
import time
from glob import glob

import numpy as np
import pandas as pd; pd.options.mode.chained_assignment = None
import torch
import yaml

from . import NISQA_lib as NL


class _Inert(object):
    """What an unknown pickle global becomes in the salvage load: a thing that can be called, built, indexed and appended
    to and does nothing -- no code of the file runs, no module is imported on its behalf."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, v):
        pass

    def __reduce__(self):
        raise TypeError('salvage stub')


def _salvage_pickle_module():
    """A ``pickle_module`` for torch.load whose Unpickler resolves ONLY the globals a tensor checkpoint needs (torch's
    tensor / storage rebuild helpers, OrderedDict, datetime, a few builtins) and turns every other global into _Inert."""
    import _codecs
    import collections
    import datetime
    import pickle
    import types
    allowed = {('collections', 'OrderedDict'): collections.OrderedDict, ('datetime', 'datetime'): datetime.datetime,
               ('_codecs', 'encode'): _codecs.encode,      # how protocol 2 spells a bytes literal (datetime's state)
               ('builtins', 'set'): set, ('builtins', 'frozenset'): frozenset, ('builtins', 'list'): list,
               ('builtins', 'dict'): dict, ('builtins', 'tuple'): tuple, ('builtins', 'int'): int, ('builtins', 'float'): float,
               ('builtins', 'bool'): bool, ('builtins', 'str'): str, ('builtins', 'bytes'): bytes,
               ('builtins', 'complex'): complex, ('builtins', 'slice'): slice, ('builtins', 'range'): range}
    for name in ('_rebuild_tensor_v2', '_rebuild_tensor', '_rebuild_parameter', '_rebuild_parameter_with_state'):
        if hasattr(torch._utils, name):
            allowed[('torch._utils', name)] = getattr(torch._utils, name)
    for name in ('FloatStorage', 'DoubleStorage', 'HalfStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage', 'ShortStorage',
                 'CharStorage', 'ByteStorage', 'BoolStorage', 'Size', 'device'):
        if hasattr(torch, name):
            allowed[('torch', name)] = getattr(torch, name)
    allowed[('torch.storage', 'UntypedStorage')] = torch.UntypedStorage
    allowed[('torch.serialization', '_get_layout')] = torch.serialization._get_layout

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            got = allowed.get((module, name))
            if got is None and module == 'torch' and isinstance(getattr(torch, name, None), torch.dtype):
                got = getattr(torch, name)
            return _Inert if got is None else got

    # EVERY pickle stream goes through the allow-listing find_class: torch's legacy (non-zip) reader pulls the magic number,
    # the protocol version and sys_info through pickle_module.load, which with the stock pickle.load is the unrestricted
    # unpickler (a file whose first pickle is a __reduce__ ran its callable here)
    def load(f, **kw):
        return Unpickler(f, **kw).load()

    def loads(b, **kw):
        import io
        return Unpickler(io.BytesIO(b), **kw).load()

    mod = types.ModuleType('nisqa_salvage_pickle')
    mod.Unpickler, mod.load, mod.loads = Unpickler, load, loads
    mod.__name__ = 'pickle'                       # torch.load consults the module's name in one branch
    return mod


def _plain(v, depth=0):
    """True if v is made of plain Python values / containers / datetime / tensors only (no salvage stub anywhere)."""
    import datetime
    if v is None or isinstance(v, (bool, int, float, str, bytes, complex, datetime.datetime, torch.Tensor, torch.Size)):
        return True
    if depth > 8:
        return False
    if isinstance(v, dict):
        return all(_plain(k, depth + 1) and _plain(x, depth + 1) for k, x in v.items())
    if isinstance(v, (list, tuple, set, frozenset)):
        return all(_plain(x, depth + 1) for x in v)
    return False


def _load_checkpoint(path):
    """torch.load(path) like the reference (NISQA_model.py:933-939), but through torch's restricted unpickler: the shipped
    checkpoints hold tensors, plain containers and one datetime in ``args`` -- allow-listed here.

    A checkpoint the reference's TRAINER wrote also carries ``db_results`` DataFrames and numpy scalars next to the tensors
    (reference NISQA_model.py:1096-1108) and the restricted unpickler rejects it.  Such a file is SALVAGED: a second pass
    with an unpickler that resolves only tensor-rebuild globals and replaces every other global by an inert stub -- no code
    of the file runs -- from which only ``args`` and ``model_state_dict`` are kept, which is all that loading needs
    (NISQA_model.py:941-942, 1023).  If a stub ended up INSIDE those two (arguments that are not plain values), the file is
    refused unless the caller opts in with NISQA_ALLOW_UNSAFE_CHECKPOINT=1 -- then it is loaded exactly as the reference
    does.  Missing / unreadable / corrupt files raise what torch.load raises."""
    import datetime
    import pickle
    try:
        with torch.serialization.safe_globals([datetime.datetime]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:                          # the restricted unpickler's refusal, nothing else
        why = str(e).split('\n')[0][:160]
        if os.environ.get('NISQA_ALLOW_UNSAFE_CHECKPOINT') == '1':
            print('nisqa_amd: {} needs the full (unsafe) unpickler, allowed by NISQA_ALLOW_UNSAFE_CHECKPOINT=1: {}'.format(
                os.path.basename(path), why))
            return torch.load(path, map_location='cpu', weights_only=False)
        try:
            ck = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_salvage_pickle_module())
            ok = isinstance(ck, dict) and isinstance(ck.get('args'), dict) and isinstance(ck.get('model_state_dict'), dict) \
                and _plain(ck['args']) and all(isinstance(k, str) and isinstance(v, torch.Tensor)
                                                for k, v in ck['model_state_dict'].items())
        except Exception:
            ok = False
        if ok:
            dropped = sorted(k for k in ck if k not in ('args', 'model_state_dict'))
            print('nisqa_amd: {} holds pickled objects beyond tensors and plain containers; loaded args and model_state_dict '
                  'only (ignored without running them: {})'.format(os.path.basename(path), ', '.join(map(str, dropped)) or '-'))
            return {'args': ck['args'], 'model_state_dict': ck['model_state_dict']}
        raise RuntimeError(
            'nisqa_amd: {} holds pickled objects beyond tensors and plain containers ({}) inside its args / model_state_dict; '
            'loading it would run code from the file.  Set NISQA_ALLOW_UNSAFE_CHECKPOINT=1 to load it the way the reference '
            'does (torch.load, full unpickler) if you trust its source.'.format(path, why)) from e


class RowCells(object):
    """'%.6f' cells of the prediction columns, filled batch by batch while the predict loop runs (NISQA_lib._predict's on_rows):
    formatting the 100 000-row table the reference prints (NISQA_model.py:79) is 0.1 s of host time that otherwise follows the
    last batch; here it happens under the next batches' transfers.  frame_to_string still cross-checks the result against pandas.

    The cells are written right-justified to the width the column will have if every value is a non-negative number below 10
    (8 characters, or the header if that is wider) and at least one cell does not end in '0' (pandas trims zeros common to the
    whole column): ``final()`` hands them out as they are when that held for the whole column -- checked per batch on the
    VALUES, vectorised, and confirmed on one formatted cell -- so that nothing per cell is left to do behind the last batch;
    otherwise ``column()`` hands out nothing and the cells are formatted from the frame as before."""

    def __init__(self, n_rows, names):
        self.names = list(names)
        self.n = int(n_rows)
        self.width = [max(len(' ' + nm), 8) for nm in self.names]
        self.cells = [[None] * self.n for _ in self.names]
        self.filled = [0] * len(self.names)
        self.regular = [True] * len(self.names)              # every value finite, >= 0 (no sign), formats to 8 characters
        self.tail = [False] * len(self.names)                # some cell confirmed not to end in '0': no common zeros to trim

    def __call__(self, ids, rows):
        ids = [int(i) for i in ids]
        v = np.asarray(rows, dtype=np.float64)               # float32 results widened like the DataFrame columns (NL:1438, 1455-1459)
        for h, col in enumerate(self.cells[:v.shape[1]]):
            x = v[:, h]
            w = self.width[h]
            if self.regular[h] and not (np.isfinite(x).all() and not np.signbit(x).any() and (x < 9.9999994).all()):
                self.regular[h] = False
            fmt = '%' + str(w) + '.6f'
            for i, c in zip(ids, x.tolist()):
                col[i] = fmt % c
            self.filled[h] += len(ids)
            if not self.tail[h] and len(ids):
                k = int(np.argmax(np.rint(x * 1e6) % 10 != 0))
                self.tail[h] = not col[ids[k]].endswith('0')

    def _complete(self, h, values):
        col = self.cells[h]
        if len(values) != self.n or self.filled[h] != self.n:   # (every row exactly once: the loop scatters each item once)
            return False
        fmt = '%' + str(self.width[h]) + '.6f'
        return all(col[i] is not None and col[i] == fmt % values[i] for i in (0, self.n // 2, self.n - 1))

    def final(self, name, values):
        """(cells already right-justified, their width) when the whole column was seen, is regular and has no common trailing
        zeros; else None"""
        if name not in self.names:
            return None
        h = self.names.index(name)
        if not (self.regular[h] and self.tail[h] and self._complete(h, values)):
            return None
        return self.cells[h], self.width[h]

    def column(self, name, values):
        """the unpadded cells of column ``name`` if every row was seen and sampled values agree with the frame's, else None"""
        if name not in self.names:
            return None
        h = self.names.index(name)
        if not self._complete(h, values) or any(c is None for c in self.cells[h]):
            return None
        return [c.lstrip(' ') for c in self.cells[h]]


def _fast_frame_lines(df, widest=None, pre=None):
    """The lines of ``df.to_string(index=False)`` for frames of plain float / integer / string columns, or None when
    the frame has anything else (pandas' rules restated: numeric headers carry a leading blank, floats are '%.6f' with
    the zeros common to the whole column trimmed, 'NaN' for missing, every cell right-justified to the column width).
    ``widest``: optional list that receives, per column, the row of its longest cell.  ``pre``: optional RowCells with
    cells formatted while the loop ran."""
    if df.shape[0] == 0 or df.shape[1] == 0 or not df.columns.is_unique or df.columns.nlevels != 1:
        return None
    cols = []
    for name in df.columns:
        col = df[name]
        kind = col.dtype.kind
        if kind == 'f':
            v = col.to_numpy(dtype=np.float64)
            a = np.abs(v[~np.isnan(v)])
            if a.size and (not np.isfinite(a).all() or (a > 1e6).any() or ((a < 1e-6) & (a > 0)).any()):
                return None                                  # pandas switches to exponent notation
            done = pre.final(name, v) if pre is not None else None
            if done is not None:                             # formatted and justified inside the loop: nothing per cell left
                cells, w = done
                if widest is not None:
                    widest.append(0)
                cols.append([(' ' + str(name)).rjust(w)] + cells)
                continue
            cells = pre.column(name, v) if pre is not None else None
            if cells is None:
                cells = ['%.6f' % x for x in v.tolist()]     # 'nan' for missing values
            cut = 0                                          # zeros every number of the column ends with
            while cut < 6 and a.size and all(c.endswith('0', 0, len(c) - cut) for c in cells if c != 'nan'):
                cut += 1
            if cut:
                tail = '0' if cut == 6 else ''
                cells = [c if c == 'nan' else c[:len(c) - cut] + tail for c in cells]
            if a.size != v.size:
                cells = ['NaN' if c == 'nan' else c for c in cells]
            head = ' ' + str(name)
        elif kind in 'iu':
            cells, head = [str(x) for x in col.tolist()], ' ' + str(name)
        elif kind == 'O':
            cells = col.tolist()
            if not all(type(x) is str for x in cells) or any('\n' in x for x in cells):
                return None
            head = str(name)
        else:
            return None
        lens = [len(c) for c in cells]
        w_cells = max(lens)
        if widest is not None:
            widest.append(lens.index(w_cells))
        w = max(len(head), w_cells)
        if min(lens) != w:
            cells = [c.rjust(w) for c in cells]
        cols.append([head.rjust(w)] + cells)
    return [' '.join(r) for r in zip(*cols)]


def frame_to_string(df, check_rows=64, pre=None):
    """``df.to_string(index=False)`` (what the reference prints, NISQA_model.py:79), an order of magnitude faster on the
    frames predict() produces: pandas needs 2.4 s for the 100 000 rows of a predict_csv run, more than the GPU needs to
    score them.  Falls back to pandas for any frame the fast path does not cover, and cross-checks itself against pandas
    on a sample of the rows that contains each column's widest cell."""
    try:
        widest = []
        lines = _fast_frame_lines(df, widest, pre)
        if lines is not None and len(df) > check_rows:
            pick = sorted(set(range(check_rows // 2)) | set(range(len(df) - check_rows // 2, len(df))) | set(widest))
            ref = df.iloc[pick].to_string(index=False).split('\n')
            if [lines[0]] + [lines[1 + i] for i in pick] != ref:
                lines = None
        if lines is not None:
            return '\n'.join(lines)
    except Exception:                                         # any surprise: the reference's own call
        pass
    return df.to_string(index=False)


class nisqaModel(object):
    """Loads the checkpoint and the dataset table; ``predict()`` returns the frame the reference returns."""

    def __init__(self, args):
        self.args = args
        if 'mode' not in self.args:
            self.args['mode'] = 'main'
        self.runinfos = {}
        self._getDevice()
        self._loadModel()
        self._loadDatasets()
        self.args['now'] = datetime.datetime.today()
        if self.args['mode'] == 'main':
            print(yaml.dump(self.args, default_flow_style=None, sort_keys=False))

    def train(self):
        """reference NISQA_model.py:41-46: _train_mos / _train_dim.  The per-batch step (train-mode forward, backward,
        Adam) runs as HIP kernels (nisqa_amd/train.py); the epoch loop around it is nisqa_amd/trainloop.py."""
        if self.args['mode'] != 'main' or not hasattr(self, 'ds_train'):
            raise NotImplementedError("train() needs the training configuration (mode 'main': run_train.py --yaml ...)")
        from . import trainloop
        trainloop.train(self)

    def _makeRunnameAndWriteYAML(self):
        from . import trainloop
        return trainloop.make_runname_and_write_yaml(self)

    def evaluate(self, mapping='first_order', do_print=True, do_plot=False):
        """reference NISQA_model.py:48-52: per-database / overall statistics of the predictions in ``ds_val.df``
        (needs ``predict()`` first and the subjective columns ``mos`` [, ``noi dis col loud``], ``db`` in the CSV)."""
        if self.args['dim'] == True:  # noqa: E712
            self._evaluate_dim(mapping=mapping, do_print=do_print, do_plot=do_plot)
        else:
            self._evaluate_mos(mapping=mapping, do_print=do_print, do_plot=do_plot)

    def _eval_one(self, label, target, mapping, do_print, do_plot, con_line_has_star=True):
        print('--> %s:' % label)
        db_results, r = NL.eval_results(self.ds_val.df, dcon=self.ds_val.df_con, target_mos=target,
                                        target_ci=target + '_ci', pred=target + '_pred', mapping=mapping,
                                        do_print=do_print, do_plot=do_plot)
        if self.ds_val.df_con is None:
            print('r_p_mean_file: {:0.2f}, rmse_mean_file: {:0.2f}'.format(r['r_p_mean_file'], r['rmse_mean_file']))
        elif con_line_has_star:
            print('r_p_mean_con: {:0.2f}, rmse_mean_con: {:0.2f}, rmse_star_map_mean_con: {:0.2f}'
                  .format(r['r_p_mean_con'], r['rmse_mean_con'], r['rmse_star_map_mean_con']))
        else:                                   # the reference's NOI line omits RMSE* (NISQA_model.py:636-638)
            print('r_p_mean_con: {:0.2f}, rmse_mean_con: {:0.2f}'.format(r['r_p_mean_con'], r['rmse_mean_con']))
        return db_results, r

    def _evaluate_mos(self, mapping='first_order', do_print=True, do_plot=False):
        """reference NISQA_model.py:572-594"""
        self.db_results, self.r = self._eval_one('MOS', 'mos', mapping, do_print, do_plot)

    def _evaluate_dim(self, mapping='first_order', do_print=True, do_plot=False):
        """reference NISQA_model.py:596-716: MOS, then noisiness, discontinuity, coloration, loudness"""
        self.db_results_val_mos, r_mos = self._eval_one('MOS', 'mos', mapping, do_print, do_plot)
        self.db_results_val_noi, r_noi = self._eval_one('NOI', 'noi', mapping, do_print, do_plot, con_line_has_star=False)
        self.db_results_val_dis, r_dis = self._eval_one('DIS', 'dis', mapping, do_print, do_plot)
        self.db_results_val_col, r_col = self._eval_one('COL', 'col', mapping, do_print, do_plot)
        self.db_results_val_loud, r_loud = self._eval_one('LOUD', 'loud', mapping, do_print, do_plot)
        self.r = dict(r_mos)
        for sfx, r in (('_noi', r_noi), ('_dis', r_dis), ('_col', r_col), ('_loud', r_loud)):
            self.r.update({k + sfx: v for k, v in r.items()})
        r_mean = 1 / 5 * (self.r['r_p_mean_con'] + self.r['r_p_mean_con_noi'] + self.r['r_p_mean_con_col']
                          + self.r['r_p_mean_con_dis'] + self.r['r_p_mean_con_loud'])
        print('\nAverage over MOS and dimensions: r_p={:0.3f}'.format(r_mean))

    def predict(self):
        """reference NISQA_model.py:54-81"""
        print('---> Predicting ...')
        # tr_parallel (nn.DataParallel in the reference, NISQA_model.py:56-57) is superseded: launch one
        # process per GPU with torchrun and the clips are sharded across ranks (nisqa_amd/dist.py).
        from . import dist as _dist
        rank, world = _dist.world()
        t0 = time.perf_counter()
        # one process: the table's cells are formatted as the rows come back (under the following batches' transfers); several ranks
        # receive the other shards' rows only in the closing all_gather, and format at the end
        dim = self.args['dim'] == True  # noqa: E712  (mirrors the reference's comparison)
        cells = RowCells(len(self.ds_val), ['mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred'] if dim else ['mos_pred']) \
            if world == 1 and os.environ.get('NISQA_FORMAT_IN_LOOP', '1') != '0' else None
        if dim:
            y_val_hat, y_val = NL.predict_dim(self.model, self.ds_val, self.args['tr_bs_val'], self.dev,
                                              num_workers=self.args['tr_num_workers'], on_rows=cells)
        else:
            y_val_hat, y_val = NL.predict_mos(self.model, self.ds_val, self.args['tr_bs_val'], self.dev,
                                              num_workers=self.args['tr_num_workers'], on_rows=cells)
        t1 = time.perf_counter()
        if self.args['output_dir']:
            self.ds_val.df['model'] = self.args['name']
            if rank == 0:
                self.ds_val.df.to_csv(os.path.join(self.args['output_dir'], 'NISQA_results.csv'), index=False)
        if rank == 0:
            print(frame_to_string(self.ds_val.df, pre=cells))
        # host seconds of this call: scoring (file list -> rows in the frame) and writing / printing the table
        self.timing = {'predict_s': t1 - t0, 'table_s': time.perf_counter() - t1}
        return self.ds_val.df

    # ---- datasets (reference NISQA_model.py:732-847) ---------------------------------------------
    def _loadDatasets(self):
        if self.args['mode'] == 'predict_file':
            self._loadDatasetsFile()
        elif self.args['mode'] == 'predict_dir':
            self._loadDatasetsFolder()
        elif self.args['mode'] == 'predict_csv':
            self._loadDatasetsCSVpredict()
        elif self.args['mode'] == 'main':
            self._loadDatasetsCSV()
        else:
            raise NotImplementedError('mode not available')

    def _loadDatasetsCSV(self):
        """Training and validation tables by database name (reference NISQA_model.py:851-926)."""
        dfile = pd.read_csv(os.path.join(self.args['data_dir'], self.args['csv_file']))
        wanted = set(self.args['csv_db_train'] + self.args['csv_db_val'])
        if not wanted.issubset(dfile.db.unique().tolist()):
            raise ValueError('Not all dbs found in csv:', wanted.difference(dfile.db.unique().tolist()))
        df_train = dfile[dfile.db.isin(self.args['csv_db_train'])].reset_index()
        df_val = dfile[dfile.db.isin(self.args['csv_db_val'])].reset_index()
        if self.args['csv_con'] is not None:
            dcon = pd.read_csv(os.path.join(self.args['data_dir'], self.args['csv_con']))
            dcon_train = dcon[dcon.db.isin(self.args['csv_db_train'])].reset_index()
            dcon_val = dcon[dcon.db.isin(self.args['csv_db_val'])].reset_index()
        else:
            dcon_train = dcon_val = None
        print('Training size: {}, Validation size: {}'.format(len(df_train), len(df_val)))
        self.ds_train = self._dataset(df_train, dcon_train, self.args['data_dir'], self.args['csv_deg'], False,
                                      mos_column=self.args['csv_mos_train'])
        self.ds_val = self._dataset(df_val, dcon_val, self.args['data_dir'], self.args['csv_deg'], False,
                                    mos_column=self.args['csv_mos_val'])
        self.runinfos['ds_train_len'] = len(self.ds_train)
        self.runinfos['ds_val_len'] = len(self.ds_val)

    def _dataset(self, df, df_con, data_dir, filename_column, to_memory, mos_column='predict_only'):
        a = self.args
        return NL.SpeechQualityDataset(
            df, df_con=df_con, data_dir=data_dir, filename_column=filename_column, mos_column=mos_column,
            seg_length=a['ms_seg_length'], max_length=a['ms_max_segments'], to_memory=to_memory,
            to_memory_workers=None, seg_hop_length=a['ms_seg_hop_length'], transform=None,
            ms_n_fft=a['ms_n_fft'], ms_hop_length=a['ms_hop_length'], ms_win_length=a['ms_win_length'],
            ms_n_mels=a['ms_n_mels'], ms_sr=a['ms_sr'], ms_fmax=a['ms_fmax'], ms_channel=a['ms_channel'],
            double_ended=a['double_ended'], dim=a['dim'], filename_column_ref=a.get('csv_ref')).bind_engine(
                lambda: self.model.engine(self.dev))

    def _loadDatasetsFolder(self):
        files = glob(os.path.join(self.args['data_dir'], '*.wav'))
        files = [os.path.basename(f) for f in files]
        df_val = pd.DataFrame(files, columns=['deg'])
        print('# files: {}'.format(len(df_val)))
        if len(df_val) == 0:
            raise ValueError('No wav files found in data_dir')
        self.ds_val = self._dataset(df_val, None, self.args['data_dir'], 'deg', None)

    def _loadDatasetsFile(self):
        data_dir = os.path.dirname(self.args['deg'])
        file_name = os.path.basename(self.args['deg'])
        df_val = pd.DataFrame([file_name], columns=['deg'])
        self.ds_val = self._dataset(df_val, None, data_dir, 'deg', None)

    def _loadDatasetsCSVpredict(self):
        csv_file_path = os.path.join(self.args['data_dir'], self.args['csv_file'])
        dfile = pd.read_csv(csv_file_path)
        if 'csv_con' in self.args:
            dcon = pd.read_csv(os.path.join(self.args['data_dir'], self.args['csv_con']))
        else:
            dcon = None
        self.ds_val = self._dataset(dfile, dcon, self.args['data_dir'], self.args['csv_deg'], False)

    # ---- model (reference NISQA_model.py:928-1030) -------------------------------------------------
    def _loadModel(self):
        if self.args['pretrained_model']:
            if os.path.isabs(self.args['pretrained_model']):
                model_path = os.path.join(self.args['pretrained_model'])
            else:
                model_path = os.path.join(os.getcwd(), self.args['pretrained_model'])
            checkpoint = _load_checkpoint(model_path)
            checkpoint['args'].update(self.args)                      # caller keys win (NISQA_model.py:941)
            self.args = checkpoint['args']
        else:
            checkpoint = None                                         # training from scratch (pretrained_model: false)

        if self.args['model'] == 'NISQA_DIM':
            self.args['dim'] = True
            self.args['csv_mos_train'] = None
            self.args['csv_mos_val'] = None
        else:
            self.args['dim'] = False
        if self.args['model'] == 'NISQA_DE':
            self.args['double_ended'] = True
        else:
            self.args['double_ended'] = False
            self.args['csv_ref'] = None

        keys = ['ms_seg_length', 'ms_n_mels', 'cnn_model', 'cnn_c_out_1', 'cnn_c_out_2', 'cnn_c_out_3',
                'cnn_kernel_size', 'cnn_dropout', 'cnn_pool_1', 'cnn_pool_2', 'cnn_pool_3', 'cnn_fc_out_h',
                'td', 'td_sa_d_model', 'td_sa_nhead', 'td_sa_pos_enc', 'td_sa_num_layers', 'td_sa_h',
                'td_sa_dropout', 'td_lstm_h', 'td_lstm_num_layers', 'td_lstm_dropout', 'td_lstm_bidirectional',
                'td_2', 'td_2_sa_d_model', 'td_2_sa_nhead', 'td_2_sa_pos_enc', 'td_2_sa_num_layers', 'td_2_sa_h',
                'td_2_sa_dropout', 'td_2_lstm_h', 'td_2_lstm_num_layers', 'td_2_lstm_dropout',
                'td_2_lstm_bidirectional', 'pool', 'pool_att_h', 'pool_att_dropout']
        self.model_args = {k: self.args[k] for k in keys}

        print('Model architecture: ' + self.args['model'])
        if self.args['model'] == 'NISQA':
            self.model = NL.NISQA(**self.model_args)
        elif self.args['model'] == 'NISQA_DIM':
            self.model = NL.NISQA_DIM(**self.model_args)
        elif self.args['model'] == 'NISQA_DE':
            raise NotImplementedError('NISQA_DE (double-ended) is out of scope of nisqa_amd')
        else:
            raise NotImplementedError('Model not available')

        if checkpoint is None:
            NL.init_parameters_(self.model)
            self.model.bind_args(self.args)
            return
        missing_keys, unexpected_keys = self.model.load_state_dict(checkpoint['model_state_dict'], strict=True)
        print('Loaded pretrained model from ' + self.args['pretrained_model'])
        if missing_keys:
            print('missing_keys:')
            print(missing_keys)
        if unexpected_keys:
            print('unexpected_keys:')
            print(unexpected_keys)
        self.model.bind_args(self.args)

    def _getDevice(self):
        """reference NISQA_model.py:1032-1051 (one GPU per process: LOCAL_RANK picks it under torchrun)"""
        def gpu():
            # more ranks than visible GPUs (a one-GPU box under torchrun with the gloo backend): ranks share devices
            return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
        if torch.cuda.is_available():
            self.dev = gpu()
        else:
            self.dev = torch.device('cpu')
        if 'tr_device' in self.args:
            if self.args['tr_device'] == 'cpu':
                self.dev = torch.device('cpu')
            elif self.args['tr_device'] == 'cuda':
                self.dev = gpu()
        print('Device: {}'.format(self.dev))
