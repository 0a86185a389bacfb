"""Where a sweep's (image, question) pairs come from: seeded synthetic pairs, or the reference's FILES.

The reference reads a VQA annotation list and images from disk, one sample per step:

  * ``vqa_dataset`` (``ALBEF_attack/dataset/vqa_dataset.py:12-67``): ``ann_file`` = json lists of
    ``{"question_id", "image", "question", "dataset", "answer"}``; ``Image.open(root/ann['image']).convert('RGB')``;
    ``pre_question`` (``dataset/utils.py:3-17``: lower-case, strip ``,.'!?"()*#:;~``, ``-`` and ``/`` to spaces, at most 50
    words at test time);
  * the test transform ``Resize((res, res), BICUBIC)`` -> ``ToTensor`` -> ``Normalize(0.5, 0.5)``
    (``dataset/__init__.py:35-39``; ``vlmo/transforms/square_transform.py:11-18``);
  * the in-tree per-question tables (``adv_attack.py:53-80``; ``vlmo_module.py:140-187``): ``right_part*.txt`` (question
    ids the victim answers correctly, one per line -- every other question is skipped, ``adv_attack.py:416``),
    ``*_ans_table*.txt`` json ``{qid: victim answer}``, ``all_correct_ans*.txt`` json ``{qid: [answers]}``,
    ``chatgpt_all_5k*.txt`` json ``{qid: [answer, declarative paraphrase(, negation)]}`` -- the inputs of the loss-mode
    decision and the MLM task (``adv_attack.py:428-558`` -> ``mlm_task.build_mlm_task``);
  * the attackable words of a question: whitespace words with exactly ONE word piece that are not stop words
    (``cal_text_attack_list``, ``adv_attack.py:222-230``).

``VqaFilePairs`` is that input side for the batched sweep: the annotation / table files are parsed once on the host, the
8-bit images of the NEXT batch are read and decoded by a small thread pool while the current batch is attacked, uploaded
as uint8 (4x fewer PCIe bytes than fp32) on the preprocessor's copy stream and resized + normalised on the device by the
Pillow-exact integer kernels of ``csrc/image.hip`` (``preprocess.ImagePreprocessor``).  ``SyntheticPairs`` is the seeded
workload of the benchmark; ``SyntheticUint8Pairs`` the same questions with seeded 8-bit camera-sized
images, so that the benchmark can put the input pipeline inside its timed region.
"""
import json
import os
import re
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import mlm_task

ANSWER_STOP_WORDS = ("on", "and", "in", "his", "her", "its")          # Adv_attack.filter, adv_attack.py:155-160

# Stand-in for ``filter_words.py`` + NLTK's English stop words (adv_attack.py:23-27; nltk's corpus cannot be downloaded
# offline): the closed-class words of VQA questions.  A caller with the real list passes ``stop_words=``.
DEFAULT_STOP_WORDS = frozenset(
    "a about above after again against all am an and any are as at be because been before being below between both but "
    "by can did do does doing down during each few for from further had has have having he her here hers herself him "
    "himself his how i if in into is it its itself just me more most my myself no nor not now of off on once only or "
    "other our ours ourselves out over own s same she should so some such t than that the their theirs them themselves "
    "then there these they this those through to too under until up very was we were what when where which while who "
    "whom why will with you your yours yourself yourselves ? .".split())


def pre_question(question, max_words=50):
    """``dataset/utils.py:3-17``."""
    q = re.sub(r"([,.'!?\"()*#:;~])", "", question.lower()).replace("-", " ").replace("/", " ").rstrip(" ")
    words = q.split(" ")
    return " ".join(words[:max_words]) if len(words) > max_words else q


def _read_json_tables(paths):
    out = {}
    for p in paths:
        with open(p) as fh:
            out.update(json.load(fh))
    return out


def _existing(directory, names):
    return [os.path.join(directory, n) for n in names if os.path.exists(os.path.join(directory, n))]


def load_tables(directory, flavor):
    """The reference's in-tree table files of ``directory`` (``adv_attack.py:53-80`` / ``vlmo_module.py:140-187``): the
    ``*_after`` twin of every file is merged in like there.  Returns ``dict(correct_list, victim_answers, clean_answers,
    paraphrases, all_correct_ans)``; a missing kind is None."""
    both = lambda stem: _existing(directory, [stem + ".txt", stem + "_after.txt"])            # noqa: E731
    right = both("right_part")
    correct = None
    if right:
        correct = []
        for p in right:
            with open(p) as fh:
                correct += [int(line.strip()) for line in fh if line.strip()]
    own = "albef_ans_table" if flavor == "albef" else "vlmo_ans_table"
    tables = dict(correct_list=correct)
    for key, stem in (("victim_answers", "vilt_ans_table_for_chatgpt"), ("clean_answers", own),
                      ("paraphrases", "chatgpt_all_5k"), ("all_correct_ans", "all_correct_ans")):
        files = both(stem)
        tables[key] = _read_json_tables(files) if files else None
    return tables


def read_image(path):
    """uint8 (H, W, 3): ``.npy`` arrays as stored, anything else through Pillow's ``Image.open(...).convert('RGB')``
    (``vqa_dataset.py:38``)."""
    if path.endswith(".npy"):
        a = np.load(path)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("{}: expected uint8 (H, W, 3), got {} {}".format(path, a.dtype, a.shape))
        return a
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class _Prefetcher:
    """Host side of the image input: decode the images of a batch on a few threads, one batch ahead of the attack."""

    def __init__(self, load_one, workers=4):
        self._load = load_one
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vqa-img")
        self._pending = {}
        self.seconds_blocked = 0.0          # time the sweep's thread waited for pixels (exposed decode time)

    def prefetch(self, key, items):
        if key not in self._pending:
            self._pending[key] = [self._pool.submit(self._load, it) for it in items]

    def take(self, key, items):
        self.prefetch(key, items)
        t0 = time.perf_counter()
        out = [f.result() for f in self._pending.pop(key)]
        self.seconds_blocked += time.perf_counter() - t0
        return out

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)


class SyntheticPairs:
    """Seeded synthetic pairs (the benchmark's workload): images U(-1, 1) drawn on the device, questions
    ``[CLS] + n words + [SEP]`` with n ~ U{4..max_words}, one sample in ``dual_every`` in dual-loss mode."""

    def __init__(self, n_samples, text_len, image_size, flavor, seed=0, joint=True, max_words=12, dual_every=0):
        from .sweep import synthetic_mlm_tasks, synthetic_questions
        self.n, self.image_size = n_samples, image_size
        self.ids, self.masks, self.attackable = synthetic_questions(n_samples, text_len, seed=seed, joint=joint,
                                                                    max_words=max_words)
        self.tasks = synthetic_mlm_tasks(self.ids, dual_every, flavor, seed=seed,
                                         max_len=text_len if flavor == "vlmo" else None)
        self.qids = list(range(n_samples))
        self.seconds_images = 0.0

    def prefetch(self, indices):
        pass

    def images(self, indices, device):
        from .sweep import synthetic_images
        return synthetic_images(indices, self.image_size, device)

    def close(self):
        pass


class PairSubset:
    """A view of some of a source's pairs (by index), itself a source: one rank's shard run on its own
    (``bench.py --emulate-world``)."""

    def __init__(self, source, indices):
        self.source, self.index = source, list(indices)
        self.n = len(self.index)
        self.ids, self.masks, self.attackable = source.ids[self.index], source.masks[self.index], source.attackable[self.index]
        self.tasks = [source.tasks[i] for i in self.index]
        self.qids = [source.qids[i] for i in self.index]
        self._t0 = (float(getattr(source, "seconds_images", 0.0)), float(getattr(source, "seconds_blocked", 0.0)))

    def prefetch(self, indices):
        self.source.prefetch([self.index[i] for i in indices])

    def images(self, indices, device):
        return self.source.images([self.index[i] for i in indices], device)

    @property
    def seconds_images(self):
        return float(getattr(self.source, "seconds_images", 0.0)) - self._t0[0]

    @property
    def seconds_blocked(self):
        return float(getattr(self.source, "seconds_blocked", 0.0)) - self._t0[1]

    def close(self):
        pass                                          # the parent source outlives its views


class _Uint8Source:
    """Shared device side of the 8-bit sources: upload + Pillow-exact resize + normalise into the batch tensor."""

    def _init_pipeline(self, image_size, workers):
        self.image_size = image_size
        self._pre = None
        self._fetch = _Prefetcher(self._load_one, workers)
        self.seconds_images = 0.0           # host time of images(): waiting for decoded pixels + launching the pipeline

    def prefetch(self, indices):
        self._fetch.prefetch(tuple(indices), list(indices))

    def images(self, indices, device):
        from ..preprocess import ImagePreprocessor
        t0 = time.perf_counter()
        device = torch.device(device)
        if self._pre is None or self._pre.device != device:
            self._pre = ImagePreprocessor(self.image_size, device)
        arrays = self._fetch.take(tuple(indices), list(indices))
        out = self._pre(arrays)
        self.seconds_images += time.perf_counter() - t0
        return out

    @property
    def seconds_blocked(self):
        return self._fetch.seconds_blocked

    def close(self):
        self._fetch.close()


class SyntheticUint8Pairs(_Uint8Source, SyntheticPairs):
    """``SyntheticPairs``' questions with seeded 8-bit (480, 640, 3) images (a camera frame's shape) that go through the
    real input pipeline: host array -> pinned upload -> ``csrc/image.hip`` resize + normalise (``_Uint8Source`` comes
    first in the bases: its ``prefetch`` / ``images`` / ``close`` are the ones that run)."""

    def __init__(self, n_samples, text_len, image_size, flavor, seed=0, joint=True, max_words=12, dual_every=0,
                 source_hw=(480, 640), workers=4):
        SyntheticPairs.__init__(self, n_samples, text_len, image_size, flavor, seed, joint, max_words, dual_every)
        self.source_hw = source_hw
        self._init_pipeline(image_size, workers)

    def _load_one(self, index):
        h, w = self.source_hw
        return np.random.RandomState(1_000_003 * int(index) + 17).randint(0, 256, (h, w, 3), dtype=np.uint8)


class VqaFilePairs(_Uint8Source):
    """The reference's file inputs for one sweep.

    ``questions``: path(s) of VQA annotation json lists (``vqa_dataset.py:13-15``).  An entry needs ``question_id``,
    ``image`` (path below ``image_root``; ``.npy`` uint8 (H, W, 3) or any Pillow format) and either ``question`` (text:
    needs ``tokenizer``, a ``wordpiece.WordPiece``) or ``words`` (pre-tokenised: one list of word-piece ids per
    whitespace word).  ``tables``: output of ``load_tables`` (or None: no filter, feature loss only).  With tables an
    entry may also carry pre-tokenised ``answer_words`` / ``correct_answers_words`` / ``paraphrase_words``."""

    def __init__(self, questions, image_root, flavor, text_len, image_size, tokenizer=None, tables=None,
                 stop_words=DEFAULT_STOP_WORDS, joint=True, workers=4, period_id=None):
        paths = [questions] if isinstance(questions, str) else list(questions)
        ann = []
        for p in paths:
            with open(p) as fh:
                ann += json.load(fh)
        tables = tables or {}
        correct = tables.get("correct_list")
        keep = set(correct) if correct is not None else None
        self.flavor, self.text_len, self.image_root = flavor, text_len, image_root
        self.skipped = 0
        rows, self.qids, self.files, self.tasks, self.questions = [], [], [], [], []
        stop = set(stop_words)
        tail = ()
        if flavor != "albef":                                 # the VLMO copy appends ' .' to the paraphrase
            tail = (period_id if period_id is not None else (tokenizer.vocab["."] if tokenizer is not None else None),)
            tail = tail if tail[0] is not None else ()
        for a in ann:
            qid = int(a["question_id"])
            if keep is not None and qid not in keep:          # adv_attack.py:416
                self.skipped += 1
                continue
            if "words" in a:
                words = [None] * len(a["words"])
                pieces = [tuple(int(t) for t in w) for w in a["words"]]
                body = [t for w in pieces for t in w] + [int(t) for t in a.get("tail", [])]
            else:
                if tokenizer is None:
                    raise ValueError("question {} is text: a tokenizer (vocab file) is needed".format(qid))
                # ALBEF: the dataset normalises the question (vqa_dataset.py:44); VLMo: the raw text is encoded whole
                # (a trailing '?' becomes a token) while the attackable words come from ``text.strip('?')``
                # (vlmo_module.py:1539,1922-1932)
                text = pre_question(a["question"]) if flavor == "albef" else a["question"]
                words, pieces = tokenizer.words(text if flavor == "albef" else text.strip("?"))
                body = [tokenizer.vocab[p] for p in tokenizer.tokenize(text)]
            body = body[:text_len - 2]
            ids = [mlm_task.CLS_ID] + body + [mlm_task.SEP_ID]
            mask = [1] * len(ids) + [0] * (text_len - len(ids))
            ids = ids + [mlm_task.PAD_ID] * (text_len - len(ids))
            att = [False] * text_len
            at = 1
            for w, p in zip(words, pieces):
                if joint and len(p) == 1 and (w is None or w not in stop) and at <= len(body):
                    att[at] = True
                at += len(p)
            rows.append((ids, mask, att))
            self.qids.append(qid)
            self.files.append(os.path.join(image_root, a["image"]))
            self.questions.append(a.get("question"))
            self.tasks.append(self._task(a, qid, tables, tokenizer, tail))
        self.n = len(rows)
        self.ids = torch.tensor([r[0] for r in rows], dtype=torch.int64).reshape(self.n, text_len)
        self.masks = torch.tensor([r[1] for r in rows], dtype=torch.int64).reshape(self.n, text_len)
        self.attackable = torch.tensor([r[2] for r in rows], dtype=torch.bool).reshape(self.n, text_len)
        self._init_pipeline(image_size, workers)

    def _task(self, a, qid, tables, tok, tail):
        """Loss mode + MLM task of one question from the tables (adv_attack.py:428-558)."""
        key = str(qid)
        para_t, victim_t, all_t = tables.get("paraphrases"), tables.get("victim_answers"), tables.get("all_correct_ans")
        if "paraphrase_words" in a:
            as_words = lambda ws: [tuple(int(t) for t in w) for w in ws]                         # noqa: E731
            answer, para = as_words(a["answer_words"]), as_words(a["paraphrase_words"])
            correct = [as_words(c) for c in a.get("correct_answers_words", [a["answer_words"]])]
            same = [c == answer for c in correct]
        elif para_t is not None and victim_t is not None and key in para_t and key in victim_t:
            if tok is None:
                raise ValueError("the answer / paraphrase tables are text: a tokenizer (vocab file) is needed")
            victim = victim_t[key]
            answer = tok.words(victim)[1]
            para = tok.words(para_t[key][1].strip("."))[1]
            all_correct = (all_t or {}).get(key, [victim])
            correct = [tok.words(c)[1] for c in all_correct]
            same = [c == victim for c in all_correct]
        else:
            return mlm_task.MlmTask(old_alg=1)
        stop = [tok.word_ids(w) for w in ANSWER_STOP_WORDS] if tok is not None else []
        return mlm_task.build_mlm_task(answer, correct, same, para, stop, self.flavor, tail=tail,
                                       max_len=self.text_len if self.flavor == "vlmo" else None)

    def _load_one(self, index):
        return read_image(self.files[index])
