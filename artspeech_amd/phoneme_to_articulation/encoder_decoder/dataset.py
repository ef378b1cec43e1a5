"""Dataset / collate API of the model-free path (reference: phoneme_to_articulation/encoder_decoder/dataset.py).

``pad_sequence_collate_fn`` (:27-65) and ``pad_sequence_transformer_collate_fn`` (:68-123) return the
very same 8- / 12-tuples (same field order, dtypes, padding values).  ``SyntheticArtSpeechDataset``
yields items in the 8-field layout of ``ArtSpeechDataset.__getitem__`` (:215-224) from a seeded
generator (the private MRI corpora are not available; SURVEY 8d describes the synthetic distribution).
``ArtSpeechDataset`` keeps the reference's constructor for real data and needs the same external
packages the reference needs (database_collector / vt_shape_gen), which are outside this hot path.
"""
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset

from ...helpers import make_padding_mask
from ...settings import DATASET_CONFIG, UNKNOWN


def _sorted_common(batch):
    sentence_numerized = [item[1] for item in batch]
    len_sentences = torch.tensor([len(s) for s in sentence_numerized], dtype=torch.int)
    len_sorted, order = len_sentences.sort(descending=True)
    tokens = pad_sequence(sentence_numerized, batch_first=True)[order]
    targets = pad_sequence([item[2] for item in batch], batch_first=True)[order]
    phonemes = [batch[i][3] for i in order]
    references = pad_sequence([item[4] for item in batch], batch_first=True)[order]
    sentence_frames = [batch[i][6] for i in order]
    sentences_ids = [batch[i][0] for i in order]
    # NOTE (as the reference, dataset.py:52-53): voicing is re-ordered BEFORE padding, the other padded
    # fields after; the result is the same ordering.
    voicing = pad_sequence([batch[i][7] for i in order], batch_first=True, padding_value=-1)
    return sentences_ids, tokens, targets, len_sorted, phonemes, references, sentence_frames, voicing


def pad_sequence_collate_fn(batch):
    """list of 8-field items -> (ids, tokens (B,T) int64, targets (B,T,A,2,N), lengths (B,) int32 sorted
    descending, phonemes, references (B,T,1,2,N), frame ids, voicing (B,T) padded with -1)."""
    return _sorted_common(batch)


def pad_sequence_transformer_collate_fn(batch):
    """The 8 fields above + float key-padding masks (0 / -inf) for source and target and (B,T,T) causal
    masks (0 on and below the diagonal, -inf above) for source and target."""
    common = _sorted_common(batch)
    lengths = common[3]
    batch_size = len(batch)
    pad = ~make_padding_mask(lengths)
    src_key_padding_mask = torch.zeros_like(pad).float()
    src_key_padding_mask[pad] = float("-inf")
    tgt_key_padding_mask = src_key_padding_mask.clone()
    max_length = int(max(lengths))
    tril = torch.tril(torch.ones(max_length, max_length))
    src_attn_mask = torch.zeros(batch_size, max_length, max_length).masked_fill(tril == 0, float("-inf"))
    tgt_attn_mask = src_attn_mask.clone()
    return (*common, src_key_padding_mask, tgt_key_padding_mask, src_attn_mask, tgt_attn_mask)


class SyntheticArtSpeechDataset(Dataset):
    """Seeded synthetic utterances: tokens uniform in [2, V) (0 = <blank> is the pad id, 1 = <unk>),
    contours U(0,1) like the normalised real data, lengths uniform in [min_len, max_len]."""

    def __init__(self, num_sentences, vocabulary, articulators, n_samples=50, min_len=20, max_len=200, seed=0,
                 database_name="artspeech2", voiced_tokens=None):
        self.vocabulary = vocabulary
        self.articulators = sorted(articulators)
        self.num_articulators = len(articulators)
        self.num_samples = n_samples
        self.dataset_config = DATASET_CONFIG[database_name]
        self.voiced_tokens = voiced_tokens or []
        self._tokens_by_id = {i: t for t, i in vocabulary.items()}
        g = torch.Generator().manual_seed(seed)
        self._lengths = torch.randint(min_len, max_len + 1, (num_sentences,), generator=g).tolist()
        self._seeds = torch.randint(0, 2 ** 31 - 1, (num_sentences,), generator=g).tolist()

    def __len__(self):
        return len(self._lengths)

    def __getitem__(self, index):
        length = self._lengths[index]
        g = torch.Generator().manual_seed(self._seeds[index])
        low = min(2, len(self.vocabulary) - 1)
        sentence_numerized = torch.randint(low, len(self.vocabulary), (length,), generator=g, dtype=torch.long)
        sentence_targets = torch.rand(length, self.num_articulators, 2, self.num_samples, generator=g)
        reference_arrays = torch.rand(length, 1, 2, self.num_samples, generator=g)
        sentence_tokens = [self._tokens_by_id.get(int(i), UNKNOWN) for i in sentence_numerized]
        voicing = torch.tensor([t in self.voiced_tokens for t in sentence_tokens], dtype=torch.float)
        critical_masks = torch.tensor([], dtype=torch.int)
        frame_ids = [f"{i:04d}" for i in range(length)]
        return (f"synthetic_{index:05d}", sentence_numerized, sentence_targets, sentence_tokens, reference_arrays,
                critical_masks, frame_ids, voicing)


class ArtSpeechDataset(Dataset):
    """Real-data dataset with the reference's constructor (dataset.py:131-156).  Walking the MRI
    corpora needs the reference's own data stack (``database_collector.DATABASE_COLLECTORS`` and
    ``phoneme_to_articulation.InputLoaderMixin`` with vt_shape_gen / vt_tools): real-data I/O is outside
    the accelerated path, so this class only adapts those objects when they are importable."""

    def __init__(self, datadir, database_name, sequences, vocabulary, articulators, n_samples=50, clip_tails=False,
                 TVs=None, voiced_tokens=None):
        try:
            from database_collector import DATABASE_COLLECTORS  # the reference's collectors
            from phoneme_to_articulation import InputLoaderMixin
        except ImportError as exc:
            raise ImportError(
                "ArtSpeechDataset reads the real-time MRI corpora through the reference's database_collector / "
                "vt_shape_gen stack, which is not part of artspeech_amd; use SyntheticArtSpeechDataset, or put the "
                "reference repository and its dependencies on PYTHONPATH") from exc
        self._loader = InputLoaderMixin
        self.vocabulary = vocabulary
        self.datadir = datadir
        self.articulators = sorted(articulators)
        self.num_articulators = len(articulators)
        self.num_samples = n_samples
        self.clip_tails = clip_tails
        self.TVs = TVs or []
        self.voiced_tokens = voiced_tokens or []
        data = DATABASE_COLLECTORS[database_name](datadir).collect_data(sequences)
        self.data = [d for d in data if d["has_all"]]
        self.dataset_config = DATASET_CONFIG[database_name]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        item = self.data[index]
        frames, refs = [], []
        for frame_id in item["frame_ids"]:
            arts = []
            for articulator in self.articulators:
                arr, ref = self._loader.prepare_articulator_array(
                    self.datadir, item["subject"], item["sequence"], frame_id, articulator, self.dataset_config,
                    clip_tails=self.clip_tails)
                arts.append(arr)
            frames.append(torch.stack(arts))
            refs.append(ref.unsqueeze(0))
        tokens = item["phonemes"]
        numerized = torch.tensor([self.vocabulary.get(t, self.vocabulary[UNKNOWN]) for t in tokens], dtype=torch.long)
        voicing = torch.tensor([t in self.voiced_tokens for t in tokens], dtype=torch.float)
        return (item["sentence_name"], numerized, torch.stack(frames).float(), tokens, torch.stack(refs).float(),
                torch.tensor([], dtype=torch.int), item["frame_ids"], voicing)


class HBMResidentDataset(Dataset):
    """The whole data set resident in HBM (288 GB per MI355X; the ArtSpeech corpora are a few GB): every utterance's tokens,
    contours, reference contours and voicing are uploaded ONCE, back to back in four flat device buffers, and a batch is
    assembled ON THE DEVICE by one gather / pad launch per field (``as_gather_pad_rows``) -- the per-step host work of the
    reference's loop (item construction, ``pad_sequence`` of 28 MB, H2D) disappears, which is what lets the kept
    ``run_epoch`` entry point feed the GPU.

        ds = HBMResidentDataset(ArtSpeechDataset(...), device)        # or any data set of 8-field items
        dl = DataLoader(ds, batch_size, shuffle, collate_fn=ds.collate, num_workers=0)

    ``__getitem__`` returns a light handle; ``collate`` returns the very tuple of ``pad_sequence_collate_fn`` (same field
    order, dtypes, padding values, sorted by decreasing length) with the four tensor fields already on the device
    (``run_epoch``'s ``.to(device)`` is then a no-op).  The transformer collate adds its masks on the host as before.
    num_workers must be 0: device tensors do not cross process boundaries."""

    def __init__(self, dataset, device):
        from ... import _lib
        self._lib = _lib
        self.device = torch.device(device)
        self.dataset_config = getattr(dataset, "dataset_config", None)
        self.articulators = getattr(dataset, "articulators", None)
        self.vocabulary = getattr(dataset, "vocabulary", None)
        self._meta, first, row = [], [], 0
        toks, tgts, refs, voic = [], [], [], []
        for i in range(len(dataset)):
            sid, numerized, targets, phonemes, reference, critical, frame_ids, voicing = dataset[i]
            n = int(numerized.shape[0])
            self._meta.append((sid, phonemes, frame_ids, n))
            first.append(row)
            row += n
            toks.append(numerized.long()); tgts.append(targets.float()); refs.append(reference.float()); voic.append(voicing.float())
        self._first = first
        self._tokens = torch.cat(toks).to(self.device)
        self._targets = torch.cat(tgts).to(self.device)
        self._references = torch.cat(refs).to(self.device)
        self._voicing = torch.cat(voic).to(self.device)

    def __len__(self):
        return len(self._meta)

    def __getitem__(self, index):
        return int(index)

    def _gather(self, src, first_dev, len_dev, B, T, pad):
        L = self._lib.lib()
        row_elems = src[0].numel() if src.dim() > 1 else 1
        out = torch.empty((B, T) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device)
        self._lib.check(L.as_gather_pad_rows(self._lib.ptr(src), self._lib.ptr(first_dev), self._lib.ptr(len_dev), B, T, row_elems,
                                             src.element_size(), float(pad), self._lib.ptr(out), self._lib.stream_ptr()),
                        "as_gather_pad_rows")
        return out

    def collate(self, indices):
        # order exactly as _sorted_common: a stable descending sort of the lengths
        lens = torch.tensor([self._meta[i][3] for i in indices], dtype=torch.int)
        len_sorted, order = lens.sort(descending=True)
        idx = [indices[int(o)] for o in order]
        B, T = len(idx), int(len_sorted[0])
        first_dev = torch.tensor([self._first[i] for i in idx], dtype=torch.int64).to(self.device, non_blocking=True)
        len_dev = len_sorted.to(self.device, non_blocking=True)
        tokens = self._gather(self._tokens, first_dev, len_dev, B, T, 0)
        targets = self._gather(self._targets, first_dev, len_dev, B, T, 0)
        references = self._gather(self._references, first_dev, len_dev, B, T, 0)
        voicing = self._gather(self._voicing, first_dev, len_dev, B, T, -1)
        return ([self._meta[i][0] for i in idx], tokens, targets, len_sorted, [self._meta[i][1] for i in idx], references,
                [self._meta[i][2] for i in idx], voicing)
