"""Device-side input pipeline: the per-sample waveform transforms of the training loader
that are pure index work -- SampleLongAudio's random crop (reference ops/transforms.py:292-309), ShuffleAudio's chunk
permutation (:256-271 -> ops/audio.py:55-67) -- and MixUp (:44-65 -> ops/audio.py:32-52), batched on the GPU, fed by
pinned, double-buffered host->device copies on a copy stream.

The reference runs `Compose([LoadAudio, SampleLongAudio, MapLabels, ShuffleAudio, MixUp, ...])` sample by sample in
DataLoader workers and collates on the host (train_2d_cnn.py:303-332).  Here the host only *plans*: for every sample
of a batch it makes the same random draws, from the same generators (`np.random`, `random`), in the same order as that
Compose would -- so with equal seeds and `num_workers=0` the batch is bit-identical -- and records them as index
tables; the raw clips (and MixUp partners) are uploaded once, and three kernels (`fsc_segments_gather` x 2,
`fsc_mixup_rows`) produce the padded `(N, Tmax, 1)` batch directly in HBM.
"""
import random

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream_ptr
from .audio import _even_slices

MAX_SEG = 256          # include/fsc_hip.h: fsc_segments_gather max_seg limit


class BatchPlan:
    """Host-side record of one batch: which raw rows to upload and what the device does with them."""

    def __init__(self):
        self.raw = []              # list of float32 waveforms to upload (primary clips first, then partners)
        self.segments = []         # per primary row: list of (src offset, length)
        self.lengths = []          # per primary row: length after crop / shuffle
        self.labels = []           # per primary row: multi-hot float32
        self.partner = []          # per primary row: index into the partner rows or -1
        self.partner_segments = []     # per partner row: [(src offset, length)]
        self.partner_lengths = []
        self.partner_labels = []
        self.partner_len_of_row = []   # per primary row: its partner's length (0 without)
        self.mix_start = []
        self.alpha = []


def _crop(size, sr, max_length):
    """SampleLongAudio (ops/transforms.py:301-307): (start, kept length); draws np.random.randint when cropping."""
    if max_length is not None and (size / sr) > max_length:
        keep = int(max_length * sr)
        start = int(np.random.randint(0, size - keep))
        return start, keep
    return 0, size


class DeviceInputPipeline:
    """`dataset[i]` must return the *loaded, un-augmented* sample: dict(audio=float32[T], sr=int, labels=float32[C])
    (a SoundDataset whose transform is Compose([LoadAudio() or SyntheticAudio(), MapLabels(class_map)]))."""

    def __init__(self, dataset, device="cuda", max_audio_length=10, p_shuffle=0.5, chunk_length=0.5, p_mixup=0.0):
        if torch.device(device).type != "cuda":
            raise _lib.FscError("DeviceInputPipeline needs a GPU device (no CPU fallback); got %r" % (device,))
        self.dataset = dataset
        self.device = torch.device(device)
        self.max_audio_length = max_audio_length
        self.p_shuffle = p_shuffle
        self.chunk_length = chunk_length
        self.p_mixup = p_mixup
        self._copy_stream = None
        self._pinned = [None, None]        # two staging buffers: batch k+1 is packed while batch k's copy drains
        self._copied = [None, None]        # event of the last copy that read each staging buffer
        self._slot = 0

    def switch_off_augmentations(self):
        """Compose.switch_off_augmentations (ops/transforms.py:362-365): the draws still happen, nothing fires."""
        self.p_shuffle = 0.0
        self.p_mixup = 0.0

    # ------------------------------------------------------------------ host: plan (random draws in Compose order)
    def plan(self, indices):
        plan = BatchPlan()
        partner_raw = []
        for i in indices:
            s = self.dataset[i]
            audio, sr = np.ascontiguousarray(s["audio"], dtype=np.float32), s["sr"]
            start, size = _crop(audio.size, sr, self.max_audio_length)               # SampleLongAudio
            segs = [(start, size)]
            if np.random.uniform() < self.p_shuffle:                                 # ShuffleAudio
                n_chunks = int((size / sr) / self.chunk_length)
                if n_chunks not in (0, 1):
                    pieces = list(_even_slices(size, n_chunks))
                    random.shuffle(pieces)
                    segs = [(start + p.start, p.stop - p.start) for p in pieces]
                    if len(segs) > MAX_SEG:
                        raise _lib.FscError("ShuffleAudio: %d chunks exceed the device table (%d)" % (len(segs), MAX_SEG))
            pidx, plen, mstart, alpha = -1, 0, 0, 0.5
            if np.random.uniform() < self.p_mixup:                                   # MixUp
                j = random.randint(0, len(self.dataset) - 1)                         # dataset.random_clean_sample()
                p = self.dataset[j]
                paudio = np.ascontiguousarray(p["audio"], dtype=np.float32)
                pstart, plen = _crop(paudio.size, p["sr"], self.max_audio_length)    # ... through clean_transform
                alpha = np.random.uniform(0.4, 0.6)                                  # mix_audio_and_labels
                if plen != size:
                    mstart = random.randint(0, max(size, plen) - 1 - min(size, plen))
                pidx = len(partner_raw)
                partner_raw.append(paudio)
                plan.partner_segments.append([(pstart, plen)])
                plan.partner_lengths.append(plen)
                plan.partner_labels.append(np.asarray(p["labels"], np.float32))
            plan.raw.append(audio)
            plan.segments.append(segs)
            plan.lengths.append(size)
            plan.labels.append(np.asarray(s["labels"], np.float32))
            plan.partner.append(pidx)
            plan.partner_len_of_row.append(plen)
            plan.mix_start.append(mstart)
            plan.alpha.append(alpha)
        plan.n_primary = len(plan.raw)
        plan.raw.extend(partner_raw)
        return plan

    # ------------------------------------------------------------------ host -> device
    def _stage(self, plan):
        """Pack the raw rows into a pinned buffer and start the async copy on the copy stream.
        Returns (device raw tensor (R, Tmax), event)."""
        rows, width = len(plan.raw), max(a.size for a in plan.raw)
        need = rows * width
        slot = self._slot
        self._slot ^= 1
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()           # the copy issued two batches ago must have left this buffer
        buf = self._pinned[slot]
        if buf is None or buf.numel() < need:
            buf = torch.empty(int(need * 1.25) + 1024, dtype=torch.float32).pin_memory()
            self._pinned[slot] = buf
        host = buf[:need].view(rows, width)
        hv = host.numpy()
        for r, a in enumerate(plan.raw):
            hv[r, :a.size] = a
            hv[r, a.size:] = 0.0
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._copy_stream):
            dev = host.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._copied[slot] = done
        return dev, done

    def _tables(self, segments, rows_base):
        n = len(segments)
        max_seg = max(len(s) for s in segments)
        count = np.zeros(n, np.int32)
        src = np.zeros((n, max_seg), np.int32)
        dst = np.zeros((n, max_seg + 1), np.int32)
        for r, segs in enumerate(segments):
            count[r] = len(segs)
            off = 0
            for k, (s0, ln) in enumerate(segs):
                src[r, k] = s0
                dst[r, k] = off
                off += ln
            dst[r, len(segs):] = off
        row = np.arange(rows_base, rows_base + n, dtype=np.int32)
        to = lambda a: torch.from_numpy(a).to(self.device, non_blocking=True)  # noqa: E731
        return to(row), to(count), to(src), to(dst), max_seg

    # ------------------------------------------------------------------ device
    def execute(self, plan, staged):
        """Run the index kernels on the CURRENT stream once the upload has landed.  -> dict(signal, labels)."""
        raw, done = staged
        torch.cuda.current_stream(self.device).wait_event(done)
        raw.record_stream(torch.cuda.current_stream(self.device))
        n = plan.n_primary
        t_a = max(plan.lengths)
        row, count, src, dst, max_seg = self._tables(plan.segments, 0)
        a = torch.empty(n, t_a, device=self.device, dtype=torch.float32)
        call("fsc_segments_gather", ptr(raw), raw.shape[1], ptr(row), ptr(count), ptr(src), ptr(dst), max_seg, ptr(a), n,
             t_a, stream_ptr())
        labels = torch.from_numpy(np.stack(plan.labels)).to(self.device, non_blocking=True)
        if not plan.partner_segments:
            return dict(signal=a.unsqueeze(-1), labels=labels)
        m = len(plan.partner_segments)
        t_b = max(plan.partner_lengths)
        prow, pcount, psrc, pdst, pmax = self._tables(plan.partner_segments, n)
        b = torch.empty(m, t_b, device=self.device, dtype=torch.float32)
        call("fsc_segments_gather", ptr(raw), raw.shape[1], ptr(prow), ptr(pcount), ptr(psrc), ptr(pdst), pmax, ptr(b), m,
             t_b, stream_ptr())
        t_out = max(t_a, t_b)
        out = torch.empty(n, t_out, device=self.device, dtype=torch.float32)
        alpha64 = np.asarray(plan.alpha, dtype=np.float64)
        i32 = lambda v: torch.from_numpy(np.asarray(v, np.int32)).to(self.device, non_blocking=True)  # noqa: E731
        f32 = lambda v: torch.from_numpy(np.asarray(v, np.float32)).to(self.device, non_blocking=True)  # noqa: E731
        plabels = torch.from_numpy(np.stack(plan.partner_labels)).to(self.device, non_blocking=True)
        lab_out = torch.empty_like(labels)
        # (named, so that every table outlives the launch: a temporary's block would be recycled by the next upload)
        partner, len_a, len_b, start = i32(plan.partner), i32(plan.lengths), i32(plan.partner_len_of_row), i32(plan.mix_start)
        al, om = f32(alpha64.astype(np.float32)), f32((1.0 - alpha64).astype(np.float32))
        call("fsc_mixup_rows", ptr(a), ptr(b), ptr(partner), ptr(len_a), ptr(len_b), ptr(start), ptr(al), ptr(om),
             ptr(out), n, t_a, t_b, t_out, ptr(labels), ptr(plabels), ptr(lab_out), labels.shape[1], stream_ptr())
        return dict(signal=out.unsqueeze(-1), labels=lab_out)

    def batch(self, indices):
        plan = self.plan(indices)
        return self.execute(plan, self._stage(plan))

    def iterate(self, batches):
        """Yield device batches for a list of index lists; batch k+1 is planned, packed and on the wire (copy stream,
        second pinned buffer) while the caller computes on batch k."""
        batches = list(batches)
        if not batches:
            return
        plan = self.plan(batches[0])
        staged = self._stage(plan)
        for nxt in batches[1:] + [None]:
            ready = self.execute(plan, staged)
            if nxt is not None:
                plan = self.plan(nxt)
                staged = self._stage(plan)
            yield ready
