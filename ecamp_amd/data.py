"""Synthetic stand-in for ContextBertDataset (ECAMP/Pre-training/module/pretrain_datasets.py:34-239): the real
dataset needs licensed MIMIC-CXR images + two CSVs that are not in the reference repo.  Emits the same batch dict
schema (pretrain_datasets.py:228-237) with the statistics SURVEY.md 8d prescribes."""
import torch
from torch.utils.data import Dataset

PAD, UNK, CLS, MASK, SEP = 0, 1, 2, 3, 4


IMG_MEAN, IMG_STD = 0.4721, 0.3037   # pretrain_datasets.py:52


def normalise_u8(image_u8):
    """uint8 [B, H, W] grayscale crops -> the f32 [B, 3, H, W] tensor the reference's transform produces (Grayscale(3) + ToTensor +
    Normalize, pretrain_datasets.py:50-52), in the same f32 arithmetic as the kernels that read the compact form."""
    x = image_u8.to(torch.float32).div(255.0).sub(IMG_MEAN).div(IMG_STD)
    return x.unsqueeze(1).expand(-1, 3, -1, -1).contiguous()


def synthetic_batch(B, S=256, img=448, vocab=30000, seed=0, device="cpu", image_u8=False):
    """image_u8: the compact image schema -- uint8 [B, img, img] grayscale crops (what the dataset's item is before ToTensor /
    Normalize) instead of the normalised f32 [B, 3, img, img]; `normalise_u8` maps one onto the other."""
    g = torch.Generator().manual_seed(1234 + seed)
    image = torch.randint(0, 256, (B, img, img), generator=g, dtype=torch.uint8) if image_u8 else torch.randn(B, 3, img, img, generator=g)
    labels = torch.randint(5, vocab, (B, S), generator=g)
    lens = torch.randint(max(2, S // 4), S + 1, (B,), generator=g)
    am = (torch.arange(S)[None, :] < lens[:, None]).long()
    labels = labels * am
    labels[:, 0] = CLS
    mask_here = (torch.rand(B, S, generator=g) < 0.5) & (am == 1)
    mask_here[:, 0] = False
    ids = labels.clone()
    ids[mask_here] = MASK
    weights = torch.ones(B, S)
    dim = torch.rand(B, S, generator=g) < 0.10
    weights[dim] = 0.05
    dcnt, mcnt, ldm = dim.sum(1).float(), mask_here.sum(1).float(), (dim & mask_here).sum(1).float()
    expand = torch.where((mcnt > 0) & (dcnt > 0), (0.95 * (dcnt - ldm) + mcnt) / (mcnt - 0.95 * ldm).clamp_min(1e-6), torch.ones(B))
    weights = torch.where(mask_here, weights * expand[:, None], weights)  # pretrain_datasets.py:177-181
    batch = dict(image=image, ids=ids, labels=labels, attention_mask=am, type_ids=torch.zeros(B, S, dtype=torch.long),
                 weights=weights, column=torch.randint(0, 3, (B,), generator=g), row=torch.randint(0, 3, (B,), generator=g))
    return {k: v.to(device) for k, v in batch.items()}


class SyntheticContextBertDataset(Dataset):
    """`len` samples of the schema above; `collate_fn` stacks WITHOUT the reference's .squeeze() (which drops the batch
    dimension at B == 1, pretrain_datasets.py:218-225)."""

    def __init__(self, length=1024, max_caption_length=256, img=448, vocab=30000, seed=0, image_u8=False):
        self.length, self.S, self.img, self.vocab, self.seed, self.image_u8 = length, max_caption_length, img, vocab, seed, image_u8

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        b = synthetic_batch(1, self.S, self.img, self.vocab, seed=self.seed * 1000003 + index, image_u8=self.image_u8)
        return {k: v[0] for k, v in b.items()}

    @staticmethod
    def collate_fn(instances):
        return {k: torch.stack([inst[k] for inst in instances]) for k in instances[0]}


class DevicePrefetcher:
    """Host -> HBM staging of the batch dict one step ahead, on its own HIP stream.

    The reference moves every batch inside `ECAMP.forward` with blocking `.cuda()` calls (model_ecamp.py:304-317): 616 MB of f32
    images per 256 pairs sit on the critical path (~10 ms at PCIe Gen5 rates).  Here batch i+1 is copied from (pinned) host memory
    while step i computes; the compute stream only waits for the copy's event.  Yields dicts of device tensors -- `forward` accepts
    them unchanged (its `.to(device)` calls become no-ops).  Used by `train_one_epoch` and by bench.py's timed region."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.timing = False    # bench.py: an event pair around every batch's copies on the copy stream
        self._timed = []       # (start event, end event, bytes)

    def copy_ms(self):
        """(ms, bytes, batches) of the host -> HBM copies bracketed since the last call (`timing` on; their events must have completed: call
        behind a synchronize).  The sum of the copies' own durations on the copy stream: what the PCIe path of this box delivers."""
        ms = sum(a.elapsed_time(b) for a, b, _ in self._timed)
        nbytes, n = sum(c for _, _, c in self._timed), len(self._timed)
        self._timed = []
        return ms, nbytes, n

    def __len__(self):
        return len(self.loader)

    NBUF = 3    # staging slots: the batch being consumed + AHEAD staged ones
    AHEAD = 2   # batches staged in front of the consumer

    def __iter__(self):
        dev = self.device
        side = torch.cuda.Stream(device=dev)
        # Fixed staging slots instead of a fresh device tensor per batch: with per-batch allocations the caching allocator decides,
        # from how far the host happens to run ahead of the GPU, whether a 616 MB block is free or has to be hipMalloc'ed inside a
        # training step (seen as sporadic 200 ms stalls in bench.py's timed region).  A slot is reused only after the step that
        # consumed its previous batch has finished on the compute stream (event wait on the copy stream: no host synchronisation).
        slots = [dict() for _ in range(self.NBUF)]
        done = [None] * self.NBUF

        def stage(batch, s):
            out, todo = {}, []
            for k, v in batch.items():
                if torch.is_tensor(v) and v.device.type == "cpu":
                    if not v.is_pinned():
                        v = v.pin_memory()
                    dst = slots[s].get(k)
                    if dst is None or dst.shape != v.shape or dst.dtype != v.dtype:   # (owned by the compute stream's pool: that is where it is read)
                        dst = slots[s][k] = torch.empty(v.shape, dtype=v.dtype, device=dev)
                        side.wait_stream(torch.cuda.current_stream(dev))
                    todo.append((dst, v))
                    out[k] = dst
                else:
                    out[k] = v
            with torch.cuda.stream(side):
                if done[s] is not None:
                    side.wait_event(done[s])
                if self.timing and todo:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(side)
                for dst, v in todo:
                    dst.copy_(v, non_blocking=True)
                ev = torch.cuda.Event(enable_timing=bool(self.timing and todo))
                ev.record(side)
                if self.timing and todo:
                    self._timed.append((t0, ev, sum(v.numel() * v.element_size() for _, v in todo)))
            return out, ev

        it = iter(self.loader)
        try:
            cur = stage(next(it), 0)
        except StopIteration:
            return
        ahead, i, nstaged = [], 0, 1     # batches staged beyond `cur` (at most AHEAD), index of `cur`, batches staged so far
        try:
            while cur is not None:
                batch, ev = cur
                torch.cuda.current_stream(dev).wait_event(ev)
                yield batch
                # Resumed: the consumer has queued the whole of step i on the compute stream and asks for batch i+1.  Only now is
                # the loader asked for more (the launch of a step never waits for the loader or for pin_memory()).  Staging runs
                # AHEAD batches in front of the consumer, so the copy of batch i+1 was issued a step ago and ran beside the GPU's
                # step i-1 .. i: the compute stream's wait above finds it finished.
                done[i % self.NBUF] = torch.cuda.Event()
                done[i % self.NBUF].record(torch.cuda.current_stream(dev))
                while len(ahead) < self.AHEAD:
                    try:
                        ahead.append(stage(next(it), nstaged % self.NBUF))   # the slot of batch nstaged - NBUF, consumed by a step whose `done` is recorded
                        nstaged += 1
                    except StopIteration:
                        break
                cur = ahead.pop(0) if ahead else None
                i += 1
        finally:
            # an abandoned iterator (exception, early break) may leave copies in flight into slots: the slots go back to the compute
            # stream's allocator pool when this frame dies, so that stream must not touch them before the copy stream is done
            torch.cuda.current_stream(dev).wait_stream(side)
