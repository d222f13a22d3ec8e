"""Torch-CPU stand-in for recommendsystem_amd.dist.HipShardKernels (test infrastructure only).

Same methods and the same buffer layouts (dist.BucketLayout: key buffers [N, S + F] with bucket (o, f) at
o*(S+F) + boff[f] + f, count slot first; row / gradient buffers [N*S (+ a zero row), W] with the d-wide row in the
first columns and the 1-dim value in column lr_col (dist.payload_width); route = position o*S + boff[f] + r), written
as plain loops, so that the host-side exchange protocol
(ShardedExchange + Comm + ShardSpec) can be rehearsed with the gloo backend on a machine without a GPU.
"""
import torch


class TorchShardKernels:
    def __init__(self, spec, B, d, with_emb, with_lr, buckets):
        from recommendsystem_amd.dist import payload_width
        self.spec, self.B, self.d, self.bk = spec, B, d, buckets
        self.F = len(spec.vocab)
        self.with_emb, self.with_lr = with_emb, with_lr
        self.W, self.lr_col = payload_width(d, with_emb, with_lr)
        self.local_off = [0]
        for v in spec.local_vocab[:-1]:
            self.local_off.append(self.local_off[-1] + v)
        self.own_grad = None
        self.own_grad1 = None

    # requester ---------------------------------------------------------------------------------
    def sort_batch(self, idx):
        F, B = idx.shape
        self.seg_keys, self.entry_seg = [], torch.full((F, B), -1, dtype=torch.int64)
        for f in range(F):
            k = idx[f]
            valid = (k > 0) & (k < self.spec.vocab[f])
            uniq, inv = torch.unique(k[valid], sorted=True, return_inverse=True)
            self.seg_keys.append(uniq)
            self.entry_seg[f, valid] = inv

    def _key_slot(self, o, f):
        return self.bk.boff[f] + f            # count slot of bucket (o, f) inside row o of a key buffer

    def route(self, send_idx, route, overflow):
        N, bk = self.spec.N, self.bk
        send_idx.zero_()
        route.fill_(-1)
        for f in range(self.F):
            keys = self.seg_keys[f]
            cap_b = bk.capf[f]
            for o in range(N):
                sel = (keys % N == o).nonzero().view(-1)
                tot = sel.numel()
                if tot > cap_b:
                    overflow.fill_(1)
                n = min(tot, cap_b)
                k0 = self._key_slot(o, f)
                send_idx[o, k0] = n
                send_idx[o, k0 + 1:k0 + 1 + n] = keys[sel[:n]] // N + 1
                route[f, sel[:n]] = (o * bk.S + bk.boff[f] + torch.arange(n)).to(route.dtype)

    def expand(self, route, pay, emb, lr1):
        F, B = self.F, self.B
        rows_f = pay[:, :self.d] if self.with_emb else None
        rows1_f = pay[:, self.lr_col] if self.with_lr else None
        for f in range(F):
            for b in range(B):
                u = int(self.entry_seg[f, b])
                code = int(route[f, u]) if u >= 0 else -1
                if code >= 0:
                    if emb is not None:
                        emb[b * F + f] = rows_f[code]
                    if lr1 is not None:
                        lr1[b * F + f] = rows1_f[code]
                else:
                    if emb is not None:
                        emb[b * F + f] = 0
                    if lr1 is not None:
                        lr1[b * F + f] = 0

    def reduce_local(self, gout, dfm, S, emb, dlr):
        """One gradient per distinct key.  Per-entry gradient of (b, f): gout[b*F+f] (+ the FM term
        dfm[b] * (S[b] - emb[b, f])); 1-dim table: dlr[b]."""
        F, B, d = self.F, self.B, self.d
        self.rowgrad = torch.zeros(F, B, d)
        self.rowgrad1 = torch.zeros(F, B)
        for f in range(F):
            for b in range(B):
                u = int(self.entry_seg[f, b])
                if u < 0:
                    continue
                if gout is not None or dfm is not None:
                    g = torch.zeros(d)
                    if gout is not None:
                        g = g + gout[b * F + f]
                    if dfm is not None:
                        g = g + dfm[b] * (S[b] - emb[b * F + f])
                    self.rowgrad[f, u] += g
                if dlr is not None:
                    self.rowgrad1[f, u] += dlr[b]

    def pack(self, route, gbuf):
        for f in range(self.F):
            for u in range(self.seg_keys[f].numel()):
                code = int(route[f, u])
                if code < 0:
                    continue
                if self.with_emb:
                    gbuf[code, :self.d] = self.rowgrad[f, u]
                if self.with_lr:
                    gbuf[code, self.lr_col] = self.rowgrad1[f, u]

    # owner -------------------------------------------------------------------------------------
    def gather(self, recv_idx, table, table1, pay):
        S = self.bk.S
        for s in range(self.spec.N):
            for f in range(self.F):
                k0, r0 = self._key_slot(s, f), s * S + self.bk.boff[f]
                n = int(recv_idx[s, k0])
                li = recv_idx[s, k0 + 1:k0 + 1 + n] + self.local_off[f]
                if self.with_emb:
                    pay[r0:r0 + n, :self.d] = table[li]
                if self.with_lr:
                    pay[r0:r0 + n, self.lr_col] = table1[li]

    def merge(self, recv_idx, gbuf, partials):
        """Dense local-table gradients (own_grad, own_grad1) = sum of the N requesters' contributions."""
        R = sum(self.spec.local_vocab)
        S = self.bk.S
        self.own_grad = torch.zeros(R, self.d)
        self.own_grad1 = torch.zeros(R)
        for s in range(self.spec.N):
            for f in range(self.F):
                k0, r0 = self._key_slot(s, f), s * S + self.bk.boff[f]
                n = int(recv_idx[s, k0])
                li = recv_idx[s, k0 + 1:k0 + 1 + n] + self.local_off[f]
                if self.with_emb:
                    self.own_grad.index_add_(0, li, gbuf[r0:r0 + n, :self.d])
                if self.with_lr:
                    self.own_grad1.index_add_(0, li, gbuf[r0:r0 + n, self.lr_col])
        partials.zero_()
        partials[0] = (self.own_grad ** 2).sum() + (self.own_grad1 ** 2).sum()

    @property
    def n_own_partials(self):
        return 4


class TorchReplicaKernels:
    """Torch-CPU stand-in for recommendsystem_amd.dist.HipReplicaKernels: the same methods on the same buffer layouts
    (keys_all [N, F, B], cnt_all [N, F], g_all [N, F, B, d], g1_all [N, F, B]); the merged gradients are left as dense
    tables `grad` / `grad1` over the stacked rows, lists added in rank order."""

    def __init__(self, vocab, N, B, d):
        self.vocab, self.N, self.B, self.d, self.F = list(vocab), N, B, d, len(vocab)
        self.off = [0]
        for v in self.vocab[:-1]:
            self.off.append(self.off[-1] + v)
        self.total = sum(self.vocab)
        self.grad = self.grad1 = None

    @property
    def n_partials(self):
        return 4

    def keys_of(self, seg_rows, out):
        out.copy_(seg_rows.long() - torch.tensor(self.off).view(self.F, 1))

    def merge(self, keys_all, cnt_all, g_all, g1_all, partials):
        self.grad = torch.zeros(self.total, self.d)
        self.grad1 = torch.zeros(self.total)
        for s in range(self.N):
            for f in range(self.F):
                n = int(cnt_all[s, f])
                keys = keys_all[s, f, :n]
                assert bool((keys[1:] > keys[:-1]).all()), "lists must be ascending and duplicate-free"
                rows = keys + self.off[f]
                if g_all is not None:
                    self.grad.index_add_(0, rows, g_all[s, f, :n])
                if g1_all is not None:
                    self.grad1.index_add_(0, rows, g1_all[s, f, :n])
        partials.zero_()
        partials[0] = (self.grad ** 2).sum() + (self.grad1 ** 2).sum()
