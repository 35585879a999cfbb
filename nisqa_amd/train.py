"""One optimiser step of the CNN-SA-AP model on the GPU (BASELINE config 5).

Host-side driver of the training operators of include/nisqa_train.h: it does what the reference does per batch at
nisqa/NISQA_model.py:131-152 (``model.train(); y_hat = model(x, n_wins); loss = biasLoss.get_loss(...);
loss.backward(); opt.step(); opt.zero_grad()``) for ``model`` = NISQA / NISQA_DIM with cnn_model=adapt, td=self_att,
pool=att -- forward in train mode (BatchNorm on batch statistics over all valid segments, the reference's dropouts),
the gradient of every parameter, BatchNorm buffer updates and the Adam update -- with every operator a HIP kernel.
PyTorch here is device memory (and the collectives of the data-parallel mode); there is no autograd, no torch.nn call
and no torch kernel in the step -- dropout masks come from a Philox kernel seeded by torch.initial_seed().

Data parallel (one process per GPU, ``torch.distributed`` initialised -- RCCL on a node): every rank runs the step on its
own clips, exactly like a replica of the reference's ``nn.DataParallel`` (NISQA_model.py:88-89: BatchNorm statistics
per replica, loss normalised by the number of labelled clips of the WHOLE batch), then the flat gradient buffer --
0.9 MB, one bucket -- is all-reduced and every rank applies the same Adam update; rank 0's BatchNorm buffers are
broadcast, as DataParallel keeps replica 0's.

Parameters live in ONE flat device buffer in "kernel layout" (conv weights as [C_out][3*3*C_in], the 384 columns of
the first Linear in [y][c] order); ``state_dict()`` / ``load_state_dict()`` convert to and from the reference's keys
and shapes, so checkpoints interoperate with the reference and with the inference engine.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import dist as _dist
from . import lib as _lib
from .engine import HipNisqa, SEG_LEN

_CONV = [(1, 16), (16, 32), (32, 64), (64, 64), (64, 64), (64, 64)]      # (C_in, C_out) of conv1..conv6
_DROP_AFTER = {2: 'cnn_d1', 3: 'cnn_d2', 4: 'cnn_d3', 5: 'cnn_d4'}       # Dropout2d sites (NISQA_lib.py:696-705)


def _ptr(t, off=0):
    return t.data_ptr() + 4 * off


def step_tables(L):
    """Index tables of one training step for clips of L[b] segments (host, numpy): [(name, array)], {grouped GEMM: tiles}.
    seg_off: exclusive prefix sum of L (int32); desc_<kind>: one row per clip for nisqa_gemm_f32 -- (a_off, b_off, c_off, M, N,
    K, lda, ldb, ldc, first 64 x 64 tile) in elements -- for the ragged attention products QK^T / PV and their four gradients
    and for the pooling reductions (NISQA_lib.py:1025-1040, 1171-1183 as batched matmuls over padded tensors in the
    reference); att_off / att_len: start and length of every softmax row in the packed [sum L^2] score buffer; pool_off /
    pool_len: the same per clip for the attention-pooling softmax."""
    L = np.asarray(L, dtype=np.int64)
    B, S = len(L), int(L.sum())
    tok = np.concatenate(([0], np.cumsum(L)))
    sq = np.concatenate(([0], np.cumsum(L * L)))
    parts = [('seg_off', tok.astype(np.int32))]
    tiles = {}

    def desc(kind, a_off, b_off, c_off, M, N, K, lda, ldb, ldc):
        z = np.zeros((B, 10), np.int64)
        for j, col in enumerate((a_off, b_off, c_off, M, N, K, lda, ldb, ldc)):
            z[:, j] = col
        nt = ((z[:, 3] + 63) // 64) * ((z[:, 4] + 63) // 64)
        z[:, 9] = np.concatenate(([0], np.cumsum(nt)[:-1]))
        parts.append(('desc_' + kind, z))
        tiles[kind] = int(nt.sum())

    t, s, b = tok[:-1], sq[:-1], np.arange(B)
    desc('qk', t * 192, t * 192, s, L, L, 64, 192, 192, L)
    desc('pv', s, t * 192, t * 64, L, 64, L, L, 192, 64)
    desc('dp', t * 64, t * 192, s, L, L, 64, 64, 192, L)
    desc('dv', s, t * 64, t * 192, L, 64, L, L, 64, 192)
    desc('dq', s, t * 192, t * 192, L, 64, L, L, 192, 192)
    desc('dk', s, t * 192, t * 192, L, 64, L, L, 192, 192)
    desc('pool', t, t * 64, b * 64, 1, 64, L, L, 64, 64)
    desc('datt', b * 64, t * 64, t, 1, L, 64, 64, 64, L)
    desc('outer', t, b * 64, t * 64, L, 64, 1, L, 64, 64)
    rows_b = np.repeat(np.arange(B), L)
    within = np.arange(S) - tok[rows_b]
    parts += [('att_off', (sq[rows_b] + within * L[rows_b]).astype(np.int64)), ('att_len', L[rows_b].astype(np.int32)),
              ('pool_off', tok[:-1].astype(np.int64)), ('pool_len', L.astype(np.int32))]
    return parts, tiles


class HipTrainer(object):
    def __init__(self, args, state_dict, device=None, lr=1e-3, precision=None):
        """precision of conv2..6 (everything else is fp32 in every mode); NISQA_HIP_TRAIN_PRECISION sets the default:
          'f32'    forward, dgrad and wgrad on exact fp32 MFMA: the reference's arithmetic;
          'mixed'  (the default until the end of round 4) forward on fp32 MFMA, dgrad and wgrad on split-bf16 MFMA (bf16 hi + lo operands, three products per term,
                   fp32 accumulation): loss, y_hat and BatchNorm buffers are those of 'f32' bit for bit, every gradient
                   stays within the same 1e-3 bound of the reference fixture (measured 7e-5);
          'bf16x6' (default) forward, dgrad and wgrad at fp32 OPERAND precision on the bf16 matrix pipe: activations, gradients and weights as
                   three exact bf16 terms, six MFMA products per term pair (csrc/train_conv.hip, TERMS = 3): held to the bounds
                   of 'f32' by the same tests, at 2.7 x its matrix-pipe rate;
          'f16x4'  (round 5) the control flow of 'bf16x6' with the forward and input-gradient convolutions on TWO f16 terms per operand and all
                   four products (csrc/train_conv.hip, FMT = F16X4: the staged tensor of a group of segments scaled by a power of two from
                   its own largest magnitude, the weights from the layer's largest |W| of the step); weight gradients as 'bf16x6'.  Held
                   to the bounds of 'f32' by the same tests; an operand may be one fp32 ulp off, hence opt-in;
          'bf16x3' the forward convolutions on split-bf16 MFMA as well (the arithmetic of the inference path's default):
                   y_hat moves by <= 1e-4.  On the random-weight fixtures that is enough to move individual gradient
                   tensors by per cent (the network's Jacobian at a random initialisation is that sensitive to its
                   activations; the backward kernels themselves agree with fp32 to 1e-5, tests/test_gpu_train.py)."""
        a = args
        self.precision = precision or os.environ.get('NISQA_HIP_TRAIN_PRECISION', 'bf16x6')
        if self.precision not in ('f32', 'mixed', 'bf16x3', 'bf16x6', 'f16x4'):
            raise ValueError('precision must be f32, mixed, bf16x3, bf16x6 or f16x4, got {}'.format(self.precision))
        if not (a.get('cnn_model') == 'adapt' and a.get('td') == 'self_att' and a.get('pool') == 'att') \
                or a.get('td_2') not in (None, 'skip') or a['model'] not in ('NISQA', 'NISQA_DIM'):
            raise NotImplementedError('HIP training step covers NISQA / NISQA_DIM with cnn_model=adapt, td=self_att, '
                                      'pool=att (config/train_nisqa_cnn_sa_ap.yaml)')
        self.eng = HipNisqa(args, state_dict, device, precision='f32')       # mel front end + geometry checks
        self.lib, self.device, self.args = self.eng.lib, self.eng.device, args
        fast, exact = self.lib.nisqa_conv3x3_gemm_bf16, self.lib.nisqa_conv3x3_gemm
        self.fused_l1 = os.environ.get('NISQA_HIP_TRAIN_FUSED_L1', '1') != '0'
        self._kchunk = int(os.environ.get('NISQA_HIP_TRAIN_KCHUNK', '128'))
        self.fused_bn_bwd = os.environ.get('NISQA_HIP_TRAIN_FUSED_BN_BWD', '1') != '0'
        self.fused_fwd_stats = os.environ.get('NISQA_HIP_TRAIN_FUSED_FWD_STATS', '1') != '0'
        self.fold_bn_wgrad = os.environ.get('NISQA_HIP_TRAIN_FOLD_BN_WGRAD', '1') != '0'
        # precision 'f32': the weight gradients of layers 2..6 segment-resident on exact fp32 MFMA (csrc/train_conv.hip,
        # nisqa_segconv_wgrad_f32) instead of the implicit GEMM with split-K atomics
        self.segconv_f32 = os.environ.get('NISQA_HIP_TRAIN_SEGCONV_F32', '1') != '0' and self.precision == 'f32'
        # ... and the fp32 FORWARD convolutions of 'f32' and 'mixed' / the fp32 input gradients of 'f32' segment-resident too
        # (nisqa_segconv_f32) instead of the implicit GEMMs
        self.segconv_f32_fwd = os.environ.get('NISQA_HIP_TRAIN_SEGCONV_F32_FWD', '1') != '0' and self.precision in ('f32', 'mixed')
        self._sc32_frags, self._sc32_bufs = {}, {}
        self._conv_fwd = fast if self.precision == 'bf16x3' else exact
        self._conv_bwd = exact if self.precision in ('f32', 'bf16x6', 'f16x4') else fast     # (layer shapes train_conv.hip does not instantiate)
        # 'bf16x6' walks the control flow of 'bf16x3' -- segment-resident kernels on uint16 fragments for forward, dgrad and wgrad --
        # through the three-term entry points
        # 'f16x4' (round 5): the control flow of 'bf16x6' with the forward and input-gradient convolutions on two f16 terms of the
        # per-group scaled tensors and all four products (nisqa_segconv_f16); the weight gradients stay on the three-term kernels
        x6, h4 = self.precision == 'bf16x6', self.precision == 'f16x4'
        self._sc_frag_bytes = self.lib.nisqa_segconv_frag_bytes_x6 if x6 else self.lib.nisqa_segconv_frag_bytes_f16 if h4 else self.lib.nisqa_segconv_frag_bytes
        self._sc_pack_many = self.lib.nisqa_segconv_pack_x6_many if x6 else self.lib.nisqa_segconv_pack_f16_many if h4 else self.lib.nisqa_segconv_pack_many
        self._sc_conv = self.lib.nisqa_segconv_bf16x6 if x6 else self.lib.nisqa_segconv_f16 if h4 else self.lib.nisqa_segconv_bf16
        # split-bf16 forward / input-gradient convolutions segment-resident (csrc/train_conv.hip) where the layer shape is
        # one of the reference configuration's; weight fragments are packed once per step (_segconv_pack)
        self.segconv = os.environ.get('NISQA_HIP_TRAIN_SEGCONV', '1') != '0' and self.precision != 'f32'
        # self-attention block + pooling heads + loss, forward and backward, as one C call of a dozen launches (csrc/train_td.hip)
        self.fused_td = os.environ.get('NISQA_HIP_TRAIN_FUSED_TD', '1') != '0'
        self._td_ws = self._td_frags = None
        self._sc_frags, self._sc_bufs = {}, {}
        self._prep_key = None
        self.lr = float(lr)
        self.n_layers = int(a['td_sa_num_layers'])
        self.heads = ['pool_layers.%d.model.' % h for h in range(5)] if a['model'] == 'NISQA_DIM' else ['pool.model.']
        if self.n_layers > 4 or len(self.heads) > 8:
            # csrc/train_td.hip is laid out for TDT_MAX_LAYERS = 4 / TDT_MAX_HEADS = 8 (the reference configurations use 2 / 5);
            # deeper stacks take the operator-by-operator path, which has no such limit
            self.fused_td = False
        self.pools = [tuple(a['cnn_pool_1']), tuple(a['cnn_pool_2']), tuple(a['cnn_pool_3'])]
        self.p_cnn, self.p_td, self.p_pool = float(a['cnn_dropout']), float(a['td_sa_dropout']), float(a['pool_att_dropout'] or 0)
        if self.p_pool:
            raise NotImplementedError('pool_att_dropout > 0 is not built (0 in every shipped config)')
        self.t = 0
        self._layout(state_dict)
        self._td_poff = self._td_param_offsets()
        self.load_state_dict(state_dict)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self._sums = torch.zeros((96 + 24 * len(self.heads), 512), dtype=torch.float64, device=self.device)
        self._cast_table, self._cast_key = None, None
        self._rng_seed, self._rng_off = int(torch.initial_seed()) & (2 ** 64 - 1), 0     # torch.manual_seed governs the masks
        self._debug = {} if os.environ.get('NISQA_HIP_TRAIN_DEBUG') == '1' else None

    # ---- parameters ------------------------------------------------------------------------------------
    def _layout(self, sd):
        keys = [k for k in sd if k.split('.')[-1] not in ('running_mean', 'running_var', 'num_batches_tracked')]
        self.keys, self.off, self.kshape = keys, {}, {}
        n = 0
        for k in keys:
            shape = tuple(sd[k].shape)
            if k.startswith('cnn.model.conv') and k.endswith('.weight'):
                shape = (shape[0], 9 * shape[1])
            self.off[k], self.kshape[k] = n, shape
            n += int(np.prod(shape))
            n = (n + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.gflat = torch.zeros_like(self.flat)
        self.P = {k: self.flat[self.off[k]:self.off[k] + int(np.prod(self.kshape[k]))].view(self.kshape[k]) for k in keys}
        self.G = {k: self.gflat[self.off[k]:self.off[k] + int(np.prod(self.kshape[k]))].view(self.kshape[k]) for k in keys}

    def _td_param_offsets(self):
        """Offsets of the self-attention / pooling parameters in the flat buffers, in the order nisqa_tdtrain_* documents."""
        pfx = 'time_dependency.model.'
        keys = [pfx + 'linear.weight', pfx + 'linear.bias', pfx + 'norm1.weight', pfx + 'norm1.bias']
        for l in range(self.n_layers):
            p = pfx + 'layers.%d.' % l
            keys += [p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', p + 'self_attn.out_proj.weight',
                     p + 'self_attn.out_proj.bias', p + 'norm1.weight', p + 'norm1.bias', p + 'linear1.weight', p + 'linear1.bias',
                     p + 'linear2.weight', p + 'linear2.bias', p + 'norm2.weight', p + 'norm2.bias']
        for hp in self.heads:
            keys += [hp + 'linear1.weight', hp + 'linear1.bias', hp + 'linear2.weight', hp + 'linear2.bias', hp + 'linear3.weight',
                     hp + 'linear3.bias']
        return np.array([self.off[k] for k in keys], dtype=np.int32)

    @staticmethod
    def _to_kernel(k, v):
        if k.startswith('cnn.model.conv') and k.endswith('.weight'):
            return v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
        if k == 'time_dependency.model.linear.weight':                         # columns c*6+y -> y*64+c
            return v.reshape(v.shape[0], 64, 6).permute(0, 2, 1).reshape(v.shape[0], 384)
        return v

    @staticmethod
    def _from_kernel(k, v, ref_shape):
        if k.startswith('cnn.model.conv') and k.endswith('.weight'):
            co, ci = ref_shape[0], ref_shape[1]
            return v.reshape(co, 3, 3, ci).permute(0, 3, 1, 2)
        if k == 'time_dependency.model.linear.weight':
            return v.reshape(v.shape[0], 6, 64).permute(0, 2, 1).reshape(v.shape[0], 384)
        return v

    def load_state_dict(self, sd):
        self._ref_shape = {k: tuple(sd[k].shape) for k in sd}
        for k in self.keys:
            v = torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k]).float()
            self.P[k].copy_(self._to_kernel(k, v).contiguous().to(self.device))
        self.bn = {}
        for i in range(1, 7):
            p = 'cnn.model.bn%d.' % i
            self.bn[i] = {'mean': torch.as_tensor(np.asarray(sd[p + 'running_mean'])).float().to(self.device).clone(),
                          'var': torch.as_tensor(np.asarray(sd[p + 'running_var'])).float().to(self.device).clone(),
                          'n': int(np.asarray(sd[p + 'num_batches_tracked'])) if p + 'num_batches_tracked' in sd else 0}

    def state_dict(self):
        """Reference keys and shapes (CPU tensors): loads into the reference's model and into HipNisqa."""
        out = {}
        for k, shape in self._ref_shape.items():
            last = k.split('.')[-1]
            if last in ('running_mean', 'running_var', 'num_batches_tracked'):
                i = int(k.split('.')[2][2:])
                out[k] = (torch.tensor(self.bn[i]['n']) if last == 'num_batches_tracked'
                          else self.bn[i]['mean' if last == 'running_mean' else 'var'].cpu().clone())
            else:
                out[k] = self._from_kernel(k, self.P[k].cpu(), shape).contiguous().clone()
        return out

    def grads(self):
        """Gradients of the last step in the reference's shapes (CPU tensors)."""
        return {k: self._from_kernel(k, self.G[k].cpu(), self._ref_shape[k]).contiguous().clone() for k in self.keys}

    # ---- thin wrappers over the C ABI ------------------------------------------------------------------------
    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _ck(self, rc, what):
        _lib.check(rc, what)

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _segconv_pack(self, geo):
        """Weight fragments of this step for every layer / direction csrc/train_conv.hip takes (mode 0 forward -- 'bf16x3'
        only --, mode 1 input gradient): ONE launch (nisqa_segconv_pack_many).  self._sc_frags[(mode, i)] is absent where the
        layer shape is not instantiated; the implicit GEMM stays there."""
        self._sc_frags, self._sc32_frags = {}, {}
        if self.segconv_f32_fwd:                                # fp32 fragments: forward ('f32', 'mixed'), input gradient ('f32')
            jobs = []
            for i in range(2, 7):
                ci, co = _CONV[i - 1]
                hi, wi = geo[i - 2][2]
                if not self.lib.nisqa_segconv_supported(hi, wi, ci, co, 0 if i == 6 else 1):
                    continue
                for mode in ((0, 1) if self.precision == 'f32' else (0,)):
                    buf = self._sc32_bufs.get((mode, i))
                    if buf is None:
                        buf = self._sc32_bufs[(mode, i)] = torch.empty(self.lib.nisqa_segconv_frag_bytes_f32(mode, ci, co) // 4,
                                                                       dtype=torch.float32, device=self.device)
                    jobs.append((mode, self.P['cnn.model.conv%d.weight' % i].data_ptr(), ci, co, buf.data_ptr()))
                    self._sc32_frags[(mode, i)] = buf
            if jobs:
                n = len(jobs)
                arr_i = lambda k: (ctypes.c_int32 * n)(*[j[k] for j in jobs])
                arr_p = lambda k: (ctypes.c_void_p * n)(*[j[k] for j in jobs])
                self._ck(self.lib.nisqa_segconv_pack_f32_many(n, arr_i(0), arr_p(1), arr_i(2), arr_i(3), arr_p(4), self._st()),
                         'nisqa_segconv_pack_f32_many')
        if not self.segconv:
            return
        jobs = []
        for i in range(2, 7):
            ci, co = _CONV[i - 1]
            hi, wi = geo[i - 2][2]
            if not self.lib.nisqa_segconv_supported(hi, wi, ci, co, 0 if i == 6 else 1):
                continue
            for mode in ((0, 1) if self.precision in ('bf16x3', 'bf16x6', 'f16x4') else (1,)):
                buf = self._sc_bufs.get((mode, i))
                if buf is None:
                    buf = self._sc_bufs[(mode, i)] = torch.empty(self._sc_frag_bytes(mode, ci, co) // 2,
                                                                 dtype=torch.int16, device=self.device)
                jobs.append((mode, self.P['cnn.model.conv%d.weight' % i].data_ptr(), ci, co, buf.data_ptr()))
                self._sc_frags[(mode, i)] = buf
        if jobs:
            n = len(jobs)
            arr_i = lambda k: (ctypes.c_int32 * n)(*[j[k] for j in jobs])
            arr_p = lambda k: (ctypes.c_void_p * n)(*[j[k] for j in jobs])
            self._ck(self._sc_pack_many(n, arr_i(0), arr_p(1), arr_i(2), arr_i(3), arr_p(4), self._st()),
                     'nisqa_segconv_pack_many')

    def _gemm(self, A, B, C, M, N, K, lda, ldb, ldc, ta=0, tb=0, ksplit=1, ao=0, bo=0, co=0, bias=None, relu=0):
        self._ck(self.lib.nisqa_gemm_f32_one(_ptr(A, ao), _ptr(B, bo), _ptr(C, co), M, N, K, lda, ldb, ldc, ta, tb, ksplit,
                                             1.0, _ptr(bias) if bias is not None else None, relu, self._st()),
                 'nisqa_gemm_f32_one')

    def _ggemm(self, kind, A, B, C, ta=0, tb=0, ao=0, bo=0, co=0):
        d, tiles = self._desc[kind]
        self._ck(self.lib.nisqa_gemm_f32(_ptr(A, ao), _ptr(B, bo), _ptr(C, co), d.data_ptr(), d.shape[0], tiles, ta, tb, 1,
                                         1.0, self._st()), 'nisqa_gemm_f32')

    def _ew(self, op, x, aux=None, bias=None, rows=None, cols=None, out=None):
        out = x if out is None else out
        rows = x.numel() // (cols or x.shape[-1]) if rows is None else rows
        cols = cols or x.shape[-1]
        self._ck(self.lib.nisqa_elementwise(op, _ptr(x), _ptr(aux) if aux is not None else None,
                                            _ptr(bias) if bias is not None else None, rows, cols, _ptr(out), self._st()),
                 'nisqa_elementwise')
        return out

    def _coldot(self, a, b, rows, c):
        s = self._sums[self._sum_i]
        self._sum_i += 1
        self._ck(self.lib.nisqa_col_dot(_ptr(a), _ptr(b), rows, c, s.data_ptr(), self._st()), 'nisqa_col_dot')
        return s

    def _ksplit(self, rows, m=64, n=64):
        """K-chunks of a weight-gradient GEMM (K = rows of the batch): enough workgroups to fill 256 CUs a few times
        over (tiles x chunks ~ 2048), chunks of at least NISQA_HIP_TRAIN_KCHUNK rows (a K-tile of 32 rows takes a workgroup
        ~1 us with its single-buffered loads: a 512-row chunk is a 16 us kernel however small the product)."""
        tiles = ((m + 63) // 64) * ((n + 63) // 64)
        return int(max(1, min(2048 // tiles, rows // self._kchunk, 4096)))

    def _linear_fwd(self, X, wk, bk, rows, n_in, n_out, relu=False):
        Y = self._new(rows, n_out)
        self._gemm(X, self.P[wk], Y, rows, n_out, n_in, n_in, n_in, n_out, tb=1, bias=self.P[bk], relu=1 if relu else 0)
        return Y

    def _linear_bwd(self, dY, X, wk, bk, rows, n_in, n_out, need_dx=True):
        s = self._coldot(dY, dY, rows, n_out)
        self._defer_cast(s, 0, n_out, self.G[bk])
        self._gemm(dY, X, self.G[wk], n_out, n_in, rows, n_out, n_in, n_in, ta=1, ksplit=self._ksplit(rows, n_out, n_in))
        if not need_dx:
            return None
        dX = self._new(rows, n_in)
        self._gemm(dY, self.P[wk], dX, rows, n_in, n_out, n_out, n_in, n_in)
        return dX

    def _ln_fwd(self, X, gk, bk, rows):
        y, xh, rs = self._new(rows, 64), self._new(rows, 64), self._new(rows)
        self._ck(self.lib.nisqa_layernorm_fwd(_ptr(X), _ptr(self.P[gk]), _ptr(self.P[bk]), rows, _ptr(y), _ptr(xh), _ptr(rs),
                                              self._st()), 'nisqa_layernorm_fwd')
        return y, xh, rs

    def _ln_bwd(self, dY, xh, rs, gk, bk, rows):
        s = self._coldot(dY, xh, rows, 64)
        self._defer_cast(s, 0, 64, self.G[bk])
        self._defer_cast(s, 64, 64, self.G[gk])
        dX = self._new(rows, 64)
        self._ck(self.lib.nisqa_layernorm_bwd(_ptr(dY), _ptr(xh), _ptr(rs), _ptr(self.P[gk]), rows, _ptr(dX), self._st()),
                 'nisqa_layernorm_bwd')
        return dX

    # ---- batch bookkeeping ---------------------------------------------------------------------------------
    def _prepare(self, n_wins):
        L = np.asarray(n_wins, dtype=np.int64)
        B, S = len(L), int(L.sum())
        tok = np.concatenate(([0], np.cumsum(L)))
        sq = np.concatenate(([0], np.cumsum(L * L)))
        self.B, self.S, self.L, self.tok, self.sq = B, S, L, tok, sq
        # index tables of the step (segment offsets, one descriptor per clip for every grouped GEMM, softmax / pooling row
        # tables): built on the host, packed into ONE page-locked buffer and sent with one asynchronous copy -- fifteen
        # pageable .to(device) calls each blocked the host behind everything queued on the stream -- and kept while the
        # batch's segment counts repeat
        key = L.tobytes()
        if key != self._prep_key or os.environ.get('NISQA_HIP_TRAIN_NO_PREP_CACHE') == '1':     # (the switch: timing of the rebuild)
            parts, tiles = step_tables(L)
            td_plan = None
            if self.fused_td:
                tparts, td_plan = self._td_plan(L)
                parts = parts + tparts
            offs, total = [], 0
            for _, a in parts:
                offs.append(total)
                total += (a.nbytes + 15) // 16 * 16
            pin = self.device.type == 'cuda'
            host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=pin)
            hv = host.numpy()
            for (_, a), o in zip(parts, offs):
                hv[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
            buf = host.to(self.device, non_blocking=pin)
            tv = {}
            for (k, a), o in zip(parts, offs):
                tv[k] = buf[o:o + a.nbytes].view({'int32': torch.int32, 'int64': torch.int64}[a.dtype.name]).view(a.shape)
            self._prep_key, self._prep_host, self._prep_buf = key, host, buf       # host stays alive until the copy has run
            self._prep_tables = (tv, tiles)
            self._td_plan_cur = td_plan
        tv, tiles = self._prep_tables
        self._tv = tv
        self.seg_off = tv['seg_off']
        self._desc = {k: (tv['desc_' + k], n) for k, n in tiles.items()}
        self.att_off, self.att_len, self.pool_off, self.pool_len = tv['att_off'], tv['att_len'], tv['pool_off'], tv['pool_len']
        self._sums.zero_()
        self._sum_i = 0
        self._casts = []
        # every dropout mask of the step in one buffer, filled by two launches (Dropout2d sites, attention / FFN sites)
        self._mask_buf, self._mask_pos = None, {}
        n_sq = int(sq[-1])
        sizes = [(k, S * c, self.p_cnn) for k, c in (('cnn_d1', 32), ('cnn_d2', 64), ('cnn_d3', 64), ('cnn_d4', 64))]
        for l in range(self.n_layers):
            sizes += [('td%d_p' % l, n_sq, self.p_td)] + [('td%d_%s' % (l, t), S * 64, self.p_td) for t in ('1', 'f', '2')]
        o = 0
        for k, n, p in sizes:
            self._mask_pos[k] = (o, n)
            o += (n + 3) // 4 * 4
        self._mask_total, self._mask_split = o, self._mask_pos['td0_p'][0] if self.n_layers else o

    def _draw_masks(self):
        self._mask_buf = self._new(self._mask_total)
        for lo, hi, p in ((0, self._mask_split, self.p_cnn), (self._mask_split, self._mask_total, self.p_td)):
            if hi > lo and p > 0:
                self._ck(self.lib.nisqa_dropout_mask(self._rng_seed, self._rng_off, p, hi - lo, _ptr(self._mask_buf, lo),
                                                     self._st()), 'nisqa_dropout_mask')
                self._rng_off += (hi - lo + 3) // 4

    def _defer_cast(self, s, lo, n, dst):
        """float64 sums s[lo:lo+n] -> float32 gradient view dst, executed by one nisqa_cast_scatter at the end of backward"""
        self._casts.append(((s.data_ptr() - self._sums.data_ptr()) // 8 + lo, (dst.data_ptr() - self.gflat.data_ptr()) // 4, n))

    def _flush_casts(self):
        if not self._casts:
            return
        key = tuple(self._casts)
        if key != self._cast_key:
            self._cast_key = key
            self._cast_table = torch.tensor(self._casts, dtype=torch.int32, device=self.device)
        self._ck(self.lib.nisqa_cast_scatter(self._sums.data_ptr(), self._cast_table.data_ptr(), len(self._casts),
                                             _ptr(self.gflat), self._st()), 'nisqa_cast_scatter')

    def _mask(self, masks, key, shape, p):
        """Dropout multipliers (0 or 1/(1-p)): explicit ``masks[key]`` (tests) or fresh Bernoulli draws."""
        if masks is not None:
            m = masks.get(key)
            return None if m is None else torch.as_tensor(m, dtype=torch.float32).reshape(shape).contiguous().to(self.device)
        if p <= 0:
            return None
        if self._mask_buf is None:
            self._draw_masks()
        o, n = self._mask_pos[key]
        return self._mask_buf[o:o + n].view(shape)

    # ---- the step ------------------------------------------------------------------------------------------
    def step_pcm(self, pcm, plan, sr, y, masks=None, bias=None):
        """pcm: float32 device tensor (clips back to back), plan: BatchPlan -- mel front end fused into the step."""
        mel, floor = self.eng.mel(pcm, plan, sr, clamp=False)
        d = plan.to(self.device)
        return self._step(mel, d['frame_off'], plan.n_wins, floor, y, masks, bias)

    def step_spec(self, specs, y, masks=None, bias=None):
        """specs: list of [48, T] dB spectrograms (the input of segment_specs) -- used by the parity tests."""
        T = np.array([s.shape[1] for s in specs], dtype=np.int64)
        hop = int(self.args['ms_seg_hop_length'])
        n_wins = np.ceil((T - (SEG_LEN - 1)) / hop).astype(np.int64)
        mel = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(s, np.float32).T for s in specs], 0))).to(self.device)
        frame_off = torch.from_numpy(np.concatenate(([0], np.cumsum(T))).astype(np.int32)).to(self.device)
        floor = torch.full((len(specs),), -3.0e38, dtype=torch.float32, device=self.device)
        return self._step(mel, frame_off, n_wins, floor, y, masks, bias)

    def _td_plan(self, L):
        """Tables of the fused self-attention block for clips of L[b] segments: padded token offsets (32 per tile), the clip of
        every tile, the prefix sum of L^2, and what nisqa_tdtrain_plan lays out (workspace / fragment sizes, the descriptors of
        the one weight-gradient GEMM and of the one column-sum launch) -> ([(name, array)], plan dict)."""
        L = np.asarray(L, dtype=np.int64)
        B, S = len(L), int(L.sum())
        tiles = (L + 31) // 32
        ptok = np.concatenate(([0], np.cumsum(tiles * 32)))
        NP = int(ptok[-1])
        H = len(self.heads)
        cap = 8 + (1 + 4 * self.n_layers + 2 * H) * 10 + (2 + 6 * self.n_layers + H) * 6
        out = np.zeros(cap, dtype=np.int64)
        self._ck(self.lib.nisqa_tdtrain_plan(B, S, NP, self.n_layers, H, self._td_poff.ctypes.data, out.ctypes.data, cap),
                 'nisqa_tdtrain_plan')
        ng, nj = int(out[2]), int(out[4])
        plan = dict(NP=NP, ws=int(out[0]), frags=int(out[1]), groups=ng, tiles=int(out[3]), jobs=nj, feat=int(out[5]),
                    dfeat=int(out[6]), yhat=int(out[7]), loss=int(out[7]) + (B * H + 3) // 4 * 4)
        parts = [('td_ptok_off', ptok.astype(np.int32)), ('td_tile_clip', np.repeat(np.arange(B), tiles).astype(np.int32)),
                 ('td_sq_off', np.concatenate(([0], np.cumsum(L * L))).astype(np.int64)),
                 ('td_desc', out[8:8 + ng * 10].copy()), ('td_jobs', out[8 + ng * 10:8 + ng * 10 + nj * 6].copy())]
        return parts, plan

    def _td_buffers(self):
        """Workspace and fragment buffer of the fused block for the current batch shape (kept while they are large enough);
        the CNN writes its features straight into the workspace."""
        pl = self._td_plan_cur
        if self._td_ws is None or self._td_ws.numel() < pl['ws']:
            self._td_ws = torch.empty(pl['ws'], dtype=torch.float32, device=self.device)
        if self._td_frags is None or self._td_frags.numel() < pl['frags']:
            self._td_frags = torch.empty(pl['frags'], dtype=torch.float32, device=self.device)
        return pl

    def _td_fused(self, y, bias, masks):
        """nisqa_tdtrain_step on the features the CNN left in the workspace -> (y_hat, loss, d loss / d feat)."""
        pl, tv = self._td_plan_cur, self._tv
        B, S, H = self.B, self.S, len(self.heads)
        yv = np.asarray(y, np.float32).reshape(B, H)
        cnt = (~np.isnan(yv)).sum(0).astype(np.float32)
        if _dist.world()[1] > 1:                       # the loss is a mean over the labelled clips of the WHOLE batch
            cnt = _dist.all_reduce_sum_(torch.from_numpy(cnt.copy())).numpy()
        inv = np.where(cnt > 0, 1.0 / np.maximum(cnt, 1), 0.0).astype(np.float32)
        # labels, 1 / count per head and the bias-mapping coefficients in ONE page-locked upload
        n_host = B * H + 8 + (B * 4 if bias is not None else 0)
        pin = self.device.type == 'cuda'
        host = torch.empty(n_host, dtype=torch.float32, pin_memory=pin)
        hv = host.numpy()
        hv[:B * H] = yv.reshape(-1)
        hv[B * H:B * H + 8] = 0
        hv[B * H:B * H + H] = inv
        if bias is not None:
            hv[B * H + 8:] = np.asarray(bias, np.float32).reshape(-1)
        dev = host.to(self.device, non_blocking=pin)
        a = _lib.TdTrainArgs()
        a.n_clips, a.n_tokens, a.n_tokens_padded, a.n_layers, a.n_heads = B, S, pl['NP'], self.n_layers, H
        a.n_wgrad_groups, a.n_wgrad_tiles, a.n_colsum_jobs = pl['groups'], pl['tiles'], pl['jobs']
        a.seg_off, a.ptok_off, a.tile_clip = self.seg_off.data_ptr(), tv['td_ptok_off'].data_ptr(), tv['td_tile_clip'].data_ptr()
        a.sq_off = tv['td_sq_off'].data_ptr()
        a.params, a.grads, a.poff = self.flat.data_ptr(), self.gflat.data_ptr(), self._td_poff.ctypes.data
        a.ws, a.frags = self._td_ws.data_ptr(), self._td_frags.data_ptr()
        a.labels, a.inv_count = dev.data_ptr(), dev.data_ptr() + 4 * B * H
        a.bias_map = dev.data_ptr() + 4 * (B * H + 8) if bias is not None else None
        n_sq = int(self.sq[-1])
        keep = [dev]
        for l in range(self.n_layers):
            for field, key, shape in ((a.mask_p, 'td%d_p' % l, (n_sq,)), (a.mask_1, 'td%d_1' % l, (S, 64)),
                                      (a.mask_f, 'td%d_f' % l, (S, 64)), (a.mask_2, 'td%d_2' % l, (S, 64))):
                m = self._mask(masks, key, shape, self.p_td)
                keep.append(m)
                field[l] = m.data_ptr() if m is not None else None
        a.wgrad_desc, a.colsum_jobs = tv['td_desc'].data_ptr(), tv['td_jobs'].data_ptr()
        self._ck(self.lib.nisqa_tdtrain_step(ctypes.byref(a), self._st()), 'nisqa_tdtrain_step')
        self._td_keep = keep                            # explicit masks / the label buffer stay alive until the step has run
        ws = self._td_ws
        # y_hat and the loss leave the persistent workspace as fresh tensors: the next step overwrites the workspace, and a
        # caller that collects losses over steps must not see later values
        y_hat = ws[pl['yhat']:pl['yhat'] + B * H].view(B, H).clone()
        loss = ws[pl['loss']:pl['loss'] + 1].clone()
        if _dist.world()[1] > 1:
            loss = _dist.all_reduce_sum_(loss)
        da = ws[pl['dfeat']:pl['dfeat'] + S * 384].view(S, 6, 64)
        return y_hat, loss, da

    def _td_unfused(self, feat, y, y_dev, bias_dev, masks):
        """The self-attention block, the pooling heads and the loss operator by operator (round-3 path, ~116 launches; kept
        behind NISQA_HIP_TRAIN_FUSED_TD=0 as the cross-check of csrc/train_td.hip) -> (y_hat, loss, d loss / d feat)."""
        L_ = self.lib
        B, S, st = self.B, self.S, self._st()
        # ================= forward: self-attention =================
        pfx = 'time_dependency.model.'
        x0 = self._linear_fwd(feat, pfx + 'linear.weight', pfx + 'linear.bias', S, 384, 64)
        x, xh0, rs0 = self._ln_fwd(x0, pfx + 'norm1.weight', pfx + 'norm1.bias', S)
        scale = 1.0 / math.sqrt(64.0)
        n_sq = int(self.sq[-1])
        td = []
        for l in range(self.n_layers):
            p = pfx + 'layers.%d.' % l
            r = {'x_in': x}
            qkv = self._linear_fwd(x, p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', S, 64, 192)
            prob = self._new(n_sq)
            self._ggemm('qk', qkv, qkv, prob, tb=1, bo=64)
            self._ck(L_.nisqa_softmax_rows_fwd(_ptr(prob), self.att_off.data_ptr(), self.att_len.data_ptr(), S, scale,
                                               _ptr(prob), st), 'nisqa_softmax_rows_fwd')
            mp = self._mask(masks, 'td%d_p' % l, (n_sq,), self.p_td)
            pd = prob if mp is None else self._ew(3, prob, aux=mp, rows=1, cols=n_sq, out=self._new(n_sq))
            ctx = self._new(S, 64)
            self._ggemm('pv', pd, qkv, ctx, bo=128)
            att = self._linear_fwd(ctx, p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias', S, 64, 64)
            m1 = self._mask(masks, 'td%d_1' % l, (S, 64), self.p_td)
            if m1 is not None:
                self._ew(3, att, aux=m1)
            r1 = self._ew(4, att, aux=x, out=self._new(S, 64))
            x1, xh1, rs1 = self._ln_fwd(r1, p + 'norm1.weight', p + 'norm1.bias', S)
            hh = self._linear_fwd(x1, p + 'linear1.weight', p + 'linear1.bias', S, 64, 64, relu=True)
            mf = self._mask(masks, 'td%d_f' % l, (S, 64), self.p_td)
            hd = hh if mf is None else self._ew(3, hh, aux=mf, out=self._new(S, 64))
            f = self._linear_fwd(hd, p + 'linear2.weight', p + 'linear2.bias', S, 64, 64)
            m2 = self._mask(masks, 'td%d_2' % l, (S, 64), self.p_td)
            if m2 is not None:
                self._ew(3, f, aux=m2)
            r2 = self._ew(4, f, aux=x1, out=self._new(S, 64))
            x, xh2, rs2 = self._ln_fwd(r2, p + 'norm2.weight', p + 'norm2.bias', S)
            r.update(qkv=qkv, prob=prob, mp=mp, pd=pd, ctx=ctx, m1=m1, x1=x1, xh1=xh1, rs1=rs1, hh=hh, mf=mf, hd=hd, m2=m2,
                     xh2=xh2, rs2=rs2)
            td.append(r)

        # ================= forward: attention pooling heads, loss =================
        H = len(self.heads)
        y_hat = self._new(B, H)
        pool = []
        for hi_, hp in enumerate(self.heads):
            u = self._linear_fwd(x, hp + 'linear1.weight', hp + 'linear1.bias', S, 64, 128, relu=True)
            sc = self._linear_fwd(u, hp + 'linear2.weight', hp + 'linear2.bias', S, 128, 1)
            att = self._new(S)
            self._ck(L_.nisqa_softmax_rows_fwd(_ptr(sc), self.pool_off.data_ptr(), self.pool_len.data_ptr(), B, 1.0,
                                               _ptr(att), st), 'nisqa_softmax_rows_fwd')
            pooled = self._new(B, 64)
            self._ggemm('pool', att, x, pooled)
            self._gemm(pooled, self.P[hp + 'linear3.weight'], y_hat, B, 1, 64, 64, 64, H, tb=1, co=hi_)
            pool.append(dict(u=u, att=att, pooled=pooled))
        b3 = torch.cat([self.P[hp + 'linear3.bias'] for hp in self.heads])
        self._ew(0, y_hat, bias=b3, rows=B, cols=H)
        loss_v = self._new(1 + H)
        dyh = self._new(B, H)
        self._ck(L_.nisqa_mse_loss(_ptr(y_hat), _ptr(y_dev), _ptr(bias_dev) if bias_dev is not None else None, B, H,
                                   _ptr(loss_v), _ptr(dyh), st), 'nisqa_mse_loss')
        loss = loss_v[:1]
        if _dist.world()[1] > 1:
            # the loss is a mean over the labelled clips of the WHOLE batch: rescale this rank's share per head
            cnt = torch.as_tensor((~np.isnan(np.asarray(y, np.float32).reshape(B, H))).sum(0), dtype=torch.float32)
            tot = _dist.all_reduce_sum_(cnt.clone())
            share = torch.where(tot > 0, cnt / tot.clamp(min=1), torch.zeros_like(cnt)).to(self.device)
            self._ew(5, dyh, bias=share, rows=B, cols=H)
            loss = _dist.all_reduce_sum_((loss_v[1:] * share).sum().reshape(1))

        # ================= backward: pooling heads =================
        s = self._coldot(dyh, dyh, B, H)
        dx = torch.zeros((S, 64), dtype=torch.float32, device=self.device)
        tmp = self._new(S, 64)
        for hi_, hp in enumerate(self.heads):
            pr = pool[hi_]
            self._defer_cast(s, hi_, 1, self.G[hp + 'linear3.bias'])
            self._gemm(dyh, pr['pooled'], self.G[hp + 'linear3.weight'], 1, 64, B, H, 64, 64, ta=1, ao=hi_)
            dpooled = self._new(B, 64)
            self._gemm(dyh, self.P[hp + 'linear3.weight'], dpooled, B, 64, 1, H, 64, 64, ao=hi_)
            datt = self._new(S)
            self._ggemm('datt', dpooled, x, datt, tb=1)
            self._ggemm('outer', pr['att'], dpooled, tmp, ta=1)
            self._ew(4, dx, aux=tmp)
            self._ck(L_.nisqa_softmax_rows_bwd(_ptr(pr['att']), _ptr(datt), self.pool_off.data_ptr(), self.pool_len.data_ptr(),
                                               B, 1.0, _ptr(datt), st), 'nisqa_softmax_rows_bwd')
            du = self._linear_bwd(datt, pr['u'], hp + 'linear2.weight', hp + 'linear2.bias', S, 128, 1)
            self._ew(2, du, aux=pr['u'])
            dxh = self._linear_bwd(du, x, hp + 'linear1.weight', hp + 'linear1.bias', S, 64, 128)
            self._ew(4, dx, aux=dxh)

        # ================= backward: self-attention layers =================
        for l in reversed(range(self.n_layers)):
            p = pfx + 'layers.%d.' % l
            r = td[l]
            dr2 = self._ln_bwd(dx, r['xh2'], r['rs2'], p + 'norm2.weight', p + 'norm2.bias', S)
            df = dr2 if r['m2'] is None else self._ew(3, dr2, aux=r['m2'], out=self._new(S, 64))
            dhd = self._linear_bwd(df, r['hd'], p + 'linear2.weight', p + 'linear2.bias', S, 64, 64)
            if r['mf'] is not None:
                self._ew(3, dhd, aux=r['mf'])
            self._ew(2, dhd, aux=r['hh'])
            dx1 = self._linear_bwd(dhd, r['x1'], p + 'linear1.weight', p + 'linear1.bias', S, 64, 64)
            self._ew(4, dx1, aux=dr2)
            dr1 = self._ln_bwd(dx1, r['xh1'], r['rs1'], p + 'norm1.weight', p + 'norm1.bias', S)
            datt = dr1 if r['m1'] is None else self._ew(3, dr1, aux=r['m1'], out=self._new(S, 64))
            dctx = self._linear_bwd(datt, r['ctx'], p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias', S, 64, 64)
            dqkv = self._new(S, 192)
            dp = self._new(n_sq)
            self._ggemm('dp', dctx, r['qkv'], dp, tb=1, bo=128)
            self._ggemm('dv', r['pd'], dctx, dqkv, ta=1, co=128)
            if r['mp'] is not None:
                self._ew(3, dp, aux=r['mp'], rows=1, cols=n_sq)
            self._ck(L_.nisqa_softmax_rows_bwd(_ptr(r['prob']), _ptr(dp), self.att_off.data_ptr(), self.att_len.data_ptr(), S,
                                               scale, _ptr(dp), st), 'nisqa_softmax_rows_bwd')
            self._ggemm('dq', dp, r['qkv'], dqkv, bo=64)
            self._ggemm('dk', dp, r['qkv'], dqkv, ta=1, co=64)
            dxin = self._linear_bwd(dqkv, r['x_in'], p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', S, 64, 192)
            dx = self._ew(4, dxin, aux=dr1)
        dx0 = self._ln_bwd(dx, xh0, rs0, pfx + 'norm1.weight', pfx + 'norm1.bias', S)
        da = self._linear_bwd(dx0, feat, pfx + 'linear.weight', pfx + 'linear.bias', S, 384, 64)        # [S][6][64]

        return y_hat, loss, da

    def _step(self, mel, frame_off, n_wins, floor, y, masks, bias):
        L_ = self.lib
        self._prepare(n_wins)
        B, S, st = self.B, self.S, self._st()
        hop = int(self.args['ms_seg_hop_length'])
        self.gflat.zero_()
        # labels (and bias coefficients) through page-locked memory: a pageable .to(device) blocks the host until the
        # previous step has drained, and the GPU then idles while this step's first launches are being issued
        def up(a, cols):
            a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(B, cols))
            if self.device.type != 'cuda':
                return torch.from_numpy(a).to(self.device)
            h = torch.empty(a.shape, dtype=torch.float32, pin_memory=True)     # caching host allocator: reuse is stream-safe
            h.numpy()[...] = a
            return h.to(self.device, non_blocking=True)
        y_dev = bias_dev = None
        if not self.fused_td:
            y_dev = up(y, len(self.heads))
            bias_dev = None if bias is None else up(bias, 4)
        else:
            pl = self._td_buffers()

        # ================= forward: AdaptCNN in train mode =================
        geo = [(48, 15, self.pools[0]), (24, 7, self.pools[1]), (12, 5, (12, 5)), (12, 5, self.pools[2]), (6, 3, (6, 3)),
               (6, 1, (6, 1))]                                               # conv output (H, W) and the pool after it
        self._segconv_pack(geo)
        cnn = []
        act = None
        for i in range(1, 7):
            ci, co = _CONV[i - 1]
            h, w, (ho, wo) = geo[i - 1]
            rows = S * h * w
            wk, bk = 'cnn.model.conv%d.weight' % i, 'cnn.model.conv%d.bias' % i
            drop = self._mask(masks, _DROP_AFTER.get(i), (S, co), self.p_cnn) if i in _DROP_AFTER else None
            if i == 6 and self.fused_td:                  # the features go straight into the fused block's workspace
                out = self._td_ws[pl['feat']:pl['feat'] + S * ho * wo * co].view(S, ho * wo, co)
            else:
                out = self._new(S, ho * wo, co)
            arg = self._new(S, ho * wo, co, dtype=torch.int32)
            mr = self._new(2 * co)
            if i == 1 and self.fused_l1:
                # layer 1 straight from the spectrogram: its 720-pixel activations are never written (csrc/train.hip,
                # "Layer 1 without its activations"): patch moments -> batch statistics -> recomputed pooling windows
                mom = self._sums[self._sum_i][:54]
                sums = self._sums[self._sum_i + 1][:32]
                self._sum_i += 2
                self._ck(L_.nisqa_conv1_moments(_ptr(mel), _ptr(frame_off), _ptr(self.seg_off), _ptr(floor), B, S, hop,
                                                mom.data_ptr(), st), 'nisqa_conv1_moments')
                self._ck(L_.nisqa_conv1_bn_act_pool_fwd(_ptr(mel), _ptr(frame_off), _ptr(self.seg_off), _ptr(floor), B, S, hop,
                                                        _ptr(self.P[wk]), _ptr(self.P[bk]), mom.data_ptr(),
                                                        _ptr(self.P['cnn.model.bn1.weight']), _ptr(self.P['cnn.model.bn1.bias']),
                                                        _ptr(self.bn[1]['mean']), _ptr(self.bn[1]['var']), sums.data_ptr(), _ptr(mr),
                                                        _ptr(drop) if drop is not None else None, _ptr(out), arg.data_ptr(), st),
                         'nisqa_conv1_bn_act_pool_fwd')
                self.bn[i]['n'] += 1
                cnn.append(dict(x=None, z=None, arg=arg, mr=mr, drop=drop, h=h, w=w, ho=ho, wo=wo, ci=ci, co=co, rows=rows, mom=mom))
                act = out
                continue
            z = self._new(rows, co)
            if i == 1:                                                         # straight from the spectrogram, no patches
                self._ck(L_.nisqa_conv1_fwd(_ptr(mel), _ptr(frame_off), _ptr(self.seg_off), _ptr(floor), B, S, hop,
                                            _ptr(self.P[wk]), _ptr(self.P[bk]), _ptr(z), st), 'nisqa_conv1_fwd')
            else:                                                              # implicit GEMM: patches gathered by the loaders
                hi, wi = geo[i - 2][2]
                fr = self._sc_frags.get((0, i))
                fr32 = self._sc32_frags.get((0, i))
                if fr is None and fr32 is not None:                              # exact fp32, segment-resident
                    sums = None
                    if self.fused_fwd_stats:
                        sums = self._sums[self._sum_i]
                        self._sum_i += 1
                    self._ck(L_.nisqa_segconv_f32(0, _ptr(act), _ptr(fr32), _ptr(z), S, hi, wi, ci, co, 0 if i == 6 else 1,
                                                  _ptr(self.P[bk]), sums.data_ptr() if sums is not None else None, st),
                             'nisqa_segconv_f32 fwd')
                elif fr is not None:
                    sums = None
                    if self.fused_fwd_stats:
                        sums = self._sums[self._sum_i]
                        self._sum_i += 1
                    self._ck(self._sc_conv(0, _ptr(act), fr.data_ptr(), _ptr(z), S, hi, wi, ci, co, 0 if i == 6 else 1,
                                                   _ptr(self.P[bk]), sums.data_ptr() if sums is not None else None, st),
                             'nisqa_segconv_bf16 fwd')
                elif self.fused_fwd_stats:                                     # sum z, sum z^2 from the convolution's epilogue
                    sums = self._sums[self._sum_i]
                    self._sum_i += 1
                    self._ck(L_.nisqa_conv3x3_fwd_stats(1 if self.precision == 'bf16x3' else 0, _ptr(act), _ptr(self.P[wk]), _ptr(z),
                                                        S, hi, wi, ci, co, 0 if i == 6 else 1, _ptr(self.P[bk]), sums.data_ptr(),
                                                        st), 'nisqa_conv3x3_fwd_stats')
                else:
                    self._ck(self._conv_fwd(0, _ptr(act), _ptr(self.P[wk]), _ptr(z), S, hi, wi, ci, co, 0 if i == 6 else 1,
                                            _ptr(self.P[bk]), 1, st), 'nisqa_conv3x3_gemm fwd')
            if i == 1 or not self.fused_fwd_stats:
                sums = self._coldot(z, z, rows, co)
            self._ck(L_.nisqa_bn_act_pool_fwd(_ptr(z), sums.data_ptr(), _ptr(self.P['cnn.model.bn%d.weight' % i]),
                                              _ptr(self.P['cnn.model.bn%d.bias' % i]), _ptr(self.bn[i]['mean']),
                                              _ptr(self.bn[i]['var']), _ptr(mr), S, h, w, co, ho, wo,
                                              _ptr(drop) if drop is not None else None, _ptr(out), arg.data_ptr(), st),
                     'nisqa_bn_act_pool_fwd')
            self.bn[i]['n'] += 1
            cnn.append(dict(x=act, z=z, arg=arg, mr=mr, drop=drop, h=h, w=w, ho=ho, wo=wo, ci=ci, co=co, rows=rows))
            act = out
        feat = act                                                             # [S][6][64] = [S][384] in (y, c) order

        # ================= self-attention block, pooling heads, loss: forward and backward =================
        if self.fused_td:
            y_hat, loss, da = self._td_fused(y, bias, masks)
        else:
            y_hat, loss, da = self._td_unfused(feat, y, y_dev, bias_dev, masks)

        # ================= backward: AdaptCNN =================
        for i in range(6, 0, -1):
            c = cnn[i - 1]
            rows, co, ci = c['rows'], c['co'], c['ci']
            g, b_ = self.P['cnn.model.bn%d.weight' % i], self.P['cnn.model.bn%d.bias' % i]
            if i == 1 and c['z'] is None:                                      # fused layer 1: sparse sums + patch moments
                acc = self._sums[self._sum_i][:176]
                self._sum_i += 1
                self._ck(L_.nisqa_conv1_bn_act_pool_bwd(_ptr(mel), _ptr(frame_off), _ptr(self.seg_off), _ptr(floor), B, S, hop,
                                                        _ptr(self.P['cnn.model.conv1.weight']), _ptr(self.P['cnn.model.conv1.bias']),
                                                        c['mom'].data_ptr(), _ptr(g), _ptr(b_), _ptr(c['mr']),
                                                        _ptr(c['drop']) if c['drop'] is not None else None, _ptr(da),
                                                        c['arg'].data_ptr(), acc.data_ptr(), _ptr(self.G['cnn.model.bn1.weight']),
                                                        _ptr(self.G['cnn.model.bn1.bias']), _ptr(self.G['cnn.model.conv1.weight']),
                                                        st), 'nisqa_conv1_bn_act_pool_bwd')
                continue                                                       # conv1.bias: exactly zero (gflat was cleared)
            dz = self._new(rows, co)
            wk, bk = 'cnn.model.conv%d.weight' % i, 'cnn.model.conv%d.bias' % i
            fold = self.fused_bn_bwd and self.fold_bn_wgrad and i > 1 and (1, i) in self._sc_frags
            seg32 = self.segconv_f32 and i > 1 and bool(L_.nisqa_segconv_supported(geo[i - 2][2][0], geo[i - 2][2][1], ci, co, 0 if i == 6 else 1))
            fold32 = seg32 and self.fused_bn_bwd and self.fold_bn_wgrad
            if fold32:
                hi, wi = geo[i - 2][2]
                s2 = self._sums[self._sum_i]
                self._sum_i += 1
                dp = _ptr(c['drop']) if c['drop'] is not None else None
                self._ck(L_.nisqa_bn_pool_bwd_sums(_ptr(da), c['arg'].data_ptr(), dp, _ptr(c['z']), _ptr(c['mr']), _ptr(g), _ptr(b_), S,
                                                   c['h'], c['w'], co, c['ho'], c['wo'], s2.data_ptr(), st), 'nisqa_bn_pool_bwd_sums')
                self._ck(L_.nisqa_segconv_wgrad_f32(_ptr(c['x']), _ptr(c['z']), _ptr(da), c['arg'].data_ptr(), dp, _ptr(c['mr']),
                                                    _ptr(g), _ptr(b_), s2.data_ptr(), _ptr(dz),
                                                    _ptr(self.G['cnn.model.bn%d.weight' % i]), _ptr(self.G['cnn.model.bn%d.bias' % i]),
                                                    _ptr(self.G[wk]), S, hi, wi, ci, co, 0 if i == 6 else 1, c['ho'], c['wo'], st),
                         'nisqa_segconv_wgrad_f32 (BatchNorm backward inside)')
                sb = None
            elif fold:
                # BatchNorm / ReLU / pool / dropout backward folded into the weight-gradient kernel's staging: the sums over the
                # pooled values first, then nisqa_segconv_wgrad_bn_bf16 computes dz on the fly (and writes it for the dgrad)
                hi, wi = geo[i - 2][2]
                s2 = self._sums[self._sum_i]
                self._sum_i += 1
                dp = _ptr(c['drop']) if c['drop'] is not None else None
                self._ck(L_.nisqa_bn_pool_bwd_sums(_ptr(da), c['arg'].data_ptr(), dp, _ptr(c['z']), _ptr(c['mr']), _ptr(g), _ptr(b_), S,
                                                   c['h'], c['w'], co, c['ho'], c['wo'], s2.data_ptr(), st), 'nisqa_bn_pool_bwd_sums')
                self._ck((L_.nisqa_segconv_wgrad_bf16x6 if self.precision in ('bf16x6', 'f16x4') else L_.nisqa_segconv_wgrad_bn_bf16)(
                    _ptr(c['x']), _ptr(c['z']), _ptr(da), c['arg'].data_ptr(), dp, _ptr(c['mr']),
                                                        _ptr(g), _ptr(b_), s2.data_ptr(), _ptr(dz),
                                                        _ptr(self.G['cnn.model.bn%d.weight' % i]), _ptr(self.G['cnn.model.bn%d.bias' % i]),
                                                        _ptr(self.G[wk]), S, hi, wi, ci, co, 0 if i == 6 else 1, c['ho'], c['wo'], st),
                         'nisqa_segconv_wgrad_bn_bf16')
                sb = None
            elif self.fused_bn_bwd:
                # reductions over the pooled values first, then ONE dense pass z -> dz (csrc/train.hip); the conv bias
                # gradient (column sums of dz) is exactly zero under train-mode BatchNorm and stays at the cleared value
                s2 = self._sums[self._sum_i]
                self._sum_i += 1
                self._ck(L_.nisqa_bn_act_pool_bwd(_ptr(da), c['arg'].data_ptr(), _ptr(c['drop']) if c['drop'] is not None else None,
                                                  _ptr(c['z']), _ptr(c['mr']), _ptr(g), _ptr(b_), S, c['h'], c['w'], co, c['ho'],
                                                  c['wo'], s2.data_ptr(), _ptr(dz), _ptr(self.G['cnn.model.bn%d.weight' % i]),
                                                  _ptr(self.G['cnn.model.bn%d.bias' % i]), st), 'nisqa_bn_act_pool_bwd')
                sb = None
            else:
                s2 = self._sums[self._sum_i]
                sb = self._sums[self._sum_i + 1]
                self._sum_i += 2
                self._ck(L_.nisqa_bn_act_pool_bwd1(_ptr(da), c['arg'].data_ptr(), _ptr(c['drop']) if c['drop'] is not None else None,
                                                   _ptr(c['z']), _ptr(c['mr']), _ptr(g), _ptr(b_), S, c['h'], c['w'], co, c['ho'],
                                                   c['wo'], _ptr(dz), s2.data_ptr(), st), 'nisqa_bn_act_pool_bwd1')
                self._ck(L_.nisqa_bn_bwd2(_ptr(dz), _ptr(c['z']), s2.data_ptr(), _ptr(c['mr']), _ptr(g), rows, co,
                                          _ptr(self.G['cnn.model.bn%d.weight' % i]), _ptr(self.G['cnn.model.bn%d.bias' % i]),
                                          sb.data_ptr(), st), 'nisqa_bn_bwd2')
            wk, bk = 'cnn.model.conv%d.weight' % i, 'cnn.model.conv%d.bias' % i
            if self._debug is not None:                                      # tools/diag_cfg5.py: intermediates of the backward pass
                self._debug['da%d' % i], self._debug['dz%d' % i], self._debug['z%d' % i] = da, dz, c['z']
            if sb is not None:
                self._defer_cast(sb, 0, co, self.G[bk])
            if i == 1:
                self._ck(L_.nisqa_conv1_wgrad(_ptr(mel), _ptr(frame_off), _ptr(self.seg_off), _ptr(floor), B, S, hop, _ptr(dz),
                                              _ptr(self.G[wk]), st), 'nisqa_conv1_wgrad')
            else:
                hi, wi = geo[i - 2][2]
                pad = 0 if i == 6 else 1
                if fold or fold32:
                    pass                                                       # the weight gradient ran with the BatchNorm backward
                elif seg32:
                    self._ck(L_.nisqa_segconv_wgrad_f32(_ptr(c['x']), None, None, None, None, None, None, None, None, _ptr(dz), None, None,
                                                        _ptr(self.G[wk]), S, hi, wi, ci, co, pad, c['ho'], c['wo'], st),
                             'nisqa_segconv_wgrad_f32')
                elif (1, i) in self._sc_frags and self.precision in ('bf16x6', 'f16x4'):
                    self._ck(L_.nisqa_segconv_wgrad_bf16x6(_ptr(c['x']), None, None, None, None, None, None, None, None, _ptr(dz), None, None,
                                                           _ptr(self.G[wk]), S, hi, wi, ci, co, pad, c['ho'], c['wo'], st),
                             'nisqa_segconv_wgrad_bf16x6')
                elif (1, i) in self._sc_frags:
                    self._ck(L_.nisqa_segconv_wgrad_bf16(_ptr(c['x']), _ptr(dz), _ptr(self.G[wk]), S, hi, wi, ci, co, pad, st),
                             'nisqa_segconv_wgrad_bf16')
                else:
                    self._ck(self._conv_bwd(2, _ptr(c['x']), _ptr(dz), _ptr(self.G[wk]), S, hi, wi, ci, co, pad, None,
                                            self._ksplit(rows, co, 9 * ci), st), 'nisqa_conv3x3_gemm wgrad')
                da = self._new(S, hi * wi, ci)
                fr = self._sc_frags.get((1, i))
                fr32 = self._sc32_frags.get((1, i))
                if fr is None and fr32 is not None:
                    self._ck(L_.nisqa_segconv_f32(1, _ptr(dz), _ptr(fr32), _ptr(da), S, hi, wi, ci, co, pad, None, None, st),
                             'nisqa_segconv_f32 dgrad')
                elif fr is not None:
                    self._ck(self._sc_conv(1, _ptr(dz), fr.data_ptr(), _ptr(da), S, hi, wi, ci, co, pad, None, None, st),
                             'nisqa_segconv_bf16 dgrad')
                else:
                    self._ck(self._conv_bwd(1, _ptr(dz), _ptr(self.P[wk]), _ptr(da), S, hi, wi, ci, co, pad, None, 1, st),
                             'nisqa_conv3x3_gemm dgrad')
            c['x'] = None

        self._flush_casts()

        # ================= data parallel: one all-reduce of the flat gradient buffer =================
        if _dist.world()[1] > 1:
            _dist.all_reduce_sum_(self.gflat)
            buf = torch.cat([torch.cat([self.bn[i]['mean'], self.bn[i]['var']]) for i in range(1, 7)])
            _dist.broadcast_(buf, 0)
            o = 0
            for i in range(1, 7):
                c = self.bn[i]['mean'].numel()
                self.bn[i]['mean'].copy_(buf[o:o + c])
                self.bn[i]['var'].copy_(buf[o + c:o + 2 * c])
                o += 2 * c

        # ================= Adam =================
        self.t += 1
        self._ck(L_.nisqa_adam_step(_ptr(self.flat), _ptr(self.gflat), _ptr(self.m), _ptr(self.v), self.flat.numel(), self.lr,
                                    self.t, st), 'nisqa_adam_step')
        self.last = {'y_hat': y_hat, 'loss': loss}
        return loss
