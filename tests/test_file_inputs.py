"""The reference's FILE inputs through the batched sweep and the entry points (SURVEY.md section 8 f3 + rows a11 / a12).

Reference: ``vqa_dataset`` (``ALBEF_attack/dataset/vqa_dataset.py:12-67``), the test transform
(``dataset/__init__.py:35-39``), the in-tree tables and the qid filter (``adv_attack.py:53-80,416``), the per-question
loss-mode / MLM-task construction from the table STRINGS (``adv_attack.py:428-558``), ``<qid>.pt`` + adversarial-text json
(``adv_attack.py:714,734-735``).

CPU: the WordPiece restatement against the ``tokenizers`` library; the table / annotation parsing; the string -> task glue
against the reference-executed task cases of ``tests/golden/text_golden.json`` (the reference's own ``test_step`` /
``evaluate`` blocks produced their ``text_ids_mlm`` / ``mlm_labels`` / ``old_alg``).  GPU: 8-bit image files -> device
resize (bit-exact vs the Pillow oracle) -> the sweep; ``entry/run.py`` on files = the same attack fed the oracle's tensors.
"""
import json
import multiprocessing
import os
import socket
import sys

import numpy as np
import pytest
import torch

from vqattack_amd.attack import dataset as ds
from vqattack_amd.attack.wordpiece import WordPiece

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def text_meta():
    with open(os.path.join(GOLD, "text_golden.json")) as fh:
        return json.load(fh)


def _vocab_file(tmp_path, vocab):
    p = str(tmp_path / "vocab.txt")
    with open(p, "w") as fh:
        fh.write("\n".join(vocab) + "\n")
    return p


def test_wordpiece_equals_the_tokenizers_library(tmp_path, text_meta):
    BertWordPieceTokenizer = pytest.importorskip("tokenizers").BertWordPieceTokenizer
    vocab = list(text_meta["vocab"]) + ["cafe", "##teria", "!"]
    path = _vocab_file(tmp_path, vocab)
    ours, lib = WordPiece(path), BertWordPieceTokenizer(path, lowercase=True)
    sentences = ["What color is the cat?", "the umbrella is red.", "is the man holding kites, balls and frisbees",
                 "Two   dogs\tsitting on the table!", "xyzzy unknownword", "café cafeteria Cafés", "the [MASK] is blue",
                 "a" * 120 + " cat", "playing-frisbee on the beach/field", "", "what's the man's hat"]
    for s in sentences:
        enc = lib.encode(s, add_special_tokens=False)
        assert ours.tokenize(s) == enc.tokens, s
        assert [ours.vocab[t] for t in ours.tokenize(s)] == enc.ids, s
    words, pieces = ours.words("Is the man holding kites")
    assert words == ["is", "the", "man", "holding", "kites"] and [len(p) for p in pieces] == [1, 1, 1, 1, 2]
    assert ours.decode_word(pieces[4]) == "kites" and (ours.pad_id, ours.cls_id, ours.sep_id, ours.mask_id) == (0, 101, 102, 103)


def test_pre_question_is_the_references():
    # dataset/utils.py:3-17
    assert ds.pre_question("What's the man's hat-color?  ") == "whats the mans hat color"
    assert ds.pre_question("Is it (really) red/blue; or: green~#*") == "is it really red blue or green"
    assert ds.pre_question(" ".join(["w"] * 60)) == " ".join(["w"] * 50)


def _write_tables(directory, flavor, cases, split=True):
    """The in-tree table files (adv_attack.py:53-80) for ``cases`` = [(qid, task case)], half of them in the ``*_after``
    twin like the reference's 2988 + 3040 split."""
    os.makedirs(directory, exist_ok=True)
    first = cases[:len(cases) // 2] if split else cases
    second = cases[len(cases) // 2:] if split else []
    own = "albef_ans_table" if flavor == "albef" else "vlmo_ans_table"
    for suffix, part in (("", first), ("_after", second)):
        if not part and suffix:
            continue
        with open(os.path.join(directory, "right_part{}.txt".format(suffix)), "w") as fh:
            fh.write("".join("{}\n".format(q) for q, _ in part))
        tables = {"vilt_ans_table_for_chatgpt": {str(q): c["vilt_ans"] for q, c in part},
                  own: {str(q): c["ans"] for q, c in part},
                  "chatgpt_all_5k": {str(q): [c["ans"], c["paraphrase"]] for q, c in part},
                  "all_correct_ans": {str(q): c["all_correct_ans"] for q, c in part}}
        for stem, table in tables.items():
            with open(os.path.join(directory, stem + suffix + ".txt"), "w") as fh:
                json.dump(table, fh)


@pytest.mark.parametrize("flavor", ["albef", "vlmo"])
def test_file_pairs_build_the_references_tasks_from_table_strings(tmp_path, text_meta, flavor):
    """question / victim answer / correct answers / paraphrase as STRINGS in the reference's file formats ->
    ``VqaFilePairs.tasks`` == what the reference's own evaluate / test_step block built from the same strings."""
    vocab = _vocab_file(tmp_path, text_meta["vocab"])
    cases = [(1000 + 7 * i, c) for i, c in enumerate(t for t in text_meta["task_cases"] if t["flavor"] == flavor)]
    assert len(cases) == 8
    tables_dir = str(tmp_path / "tables")
    _write_tables(tables_dir, flavor, cases)
    ann = [{"question_id": q, "image": "{}.npy".format(q), "question": c["question"], "dataset": "vqa", "answer": [c["ans"]]}
           for q, c in cases]
    ann.insert(3, {"question_id": 5, "image": "5.npy", "question": "is the cat red", "dataset": "vqa"})   # not in right_part
    qfile = str(tmp_path / "vqa_val.json")
    with open(qfile, "w") as fh:
        json.dump(ann, fh)
    tables = ds.load_tables(tables_dir, flavor)
    assert tables["correct_list"] == [q for q, _ in cases] and len(tables["paraphrases"]) == 8      # both twins merged
    text_len = 25 if flavor == "albef" else 40
    src = ds.VqaFilePairs(qfile, str(tmp_path), flavor, text_len, 32, tokenizer=WordPiece(vocab), tables=tables,
                          stop_words=set(text_meta["filter_in_vocab"]))
    assert src.n == 8 and src.skipped == 1 and src.qids == [q for q, _ in cases]
    seen = {0: 0, 1: 0}
    for task, (q, c) in zip(src.tasks, cases):
        assert task.old_alg == c["old_alg"], c["paraphrase"]
        seen[task.old_alg] += 1
        if task.old_alg == 1:
            continue
        assert [task.text_ids_mlm] == c["text_ids_mlm"] and [task.text_mask_mlm] == c["text_mask_mlm"]
        want = c["mlm_labels"]
        assert task.mlm_labels == want[0] or [task.mlm_labels] == want or task.mlm_labels == [want[0]], c["paraphrase"]
    assert seen[0] >= 5 and seen[1] >= 1
    src.close()


def test_file_pairs_questions_and_attackable_words(tmp_path, text_meta):
    """ids = [CLS] pieces [SEP] + padding; attackable = one-piece words that are not stop words, at their piece position
    + 1 (``cal_text_attack_list``, adv_attack.py:222-230) -- checked on the reference-executed candidate cases."""
    vocab = _vocab_file(tmp_path, text_meta["vocab"])
    cases = [c for c in text_meta["cand_cases"] if c["flavor"] == "vlmo"]
    ann = [{"question_id": i, "image": "x.npy", "question": c["text"]} for i, c in enumerate(cases)]
    qfile = str(tmp_path / "q.json")
    with open(qfile, "w") as fh:
        json.dump(ann, fh)
    src = ds.VqaFilePairs(qfile, str(tmp_path), "vlmo", 40, 32, tokenizer=WordPiece(vocab),
                          stop_words=set(text_meta["filter_in_vocab"]))
    assert src.n == len(cases) and all(t.old_alg == 1 for t in src.tasks)           # no tables: feature loss only
    for i, c in enumerate(cases):
        # the question is encoded whole (its trailing '?' is a token, vlmo_module.py:1922-1928); the candidate MLM's input
        # of the fixture is the same without it (``ori_text.strip('?')``, :1539)
        qmark = WordPiece(vocab).vocab["?"]
        want_ids = c["mlm_input_ids"][:-1] + [qmark] * c["text"].endswith("?") + [102]
        n = len(want_ids)
        assert src.ids[i, :n].tolist() == want_ids and not bool(src.ids[i, n:].any())
        assert src.masks[i].tolist() == [1] * n + [0] * (40 - n)
        want = [k[0] + 1 for k, f in zip(c["keys"], c["word_filtered"]) if k[1] - k[0] == 1 and not f]
        assert torch.nonzero(src.attackable[i]).flatten().tolist() == want, c["text"]
    # pre-tokenised entries need no vocabulary
    ann2 = [{"question_id": 9, "image": "x.npy", "words": [[2054], [3609, 2015], [2003]]}]
    with open(qfile, "w") as fh:
        json.dump(ann2, fh)
    src2 = ds.VqaFilePairs(qfile, str(tmp_path), "albef", 12, 32)
    assert src2.ids[0].tolist() == [101, 2054, 3609, 2015, 2003, 102] + [0] * 6
    assert torch.nonzero(src2.attackable[0]).flatten().tolist() == [1, 4]
    with pytest.raises(ValueError):
        ds.VqaFilePairs(str(tmp_path / "q_text.json") if False else _text_only(tmp_path), str(tmp_path), "albef", 12, 32)
    src.close(), src2.close()


def _text_only(tmp_path):
    p = str(tmp_path / "text_only.json")
    with open(p, "w") as fh:
        json.dump([{"question_id": 1, "image": "x.npy", "question": "what is this"}], fh)
    return p


def test_read_image_formats(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    a = np.random.RandomState(0).randint(0, 256, (11, 13, 3)).astype(np.uint8)
    np.save(str(tmp_path / "a.npy"), a)
    Image.fromarray(a, "RGB").save(str(tmp_path / "a.png"))
    Image.fromarray(a[:, :, 0], "L").save(str(tmp_path / "grey.png"))
    assert np.array_equal(ds.read_image(str(tmp_path / "a.npy")), a)
    assert np.array_equal(ds.read_image(str(tmp_path / "a.png")), a)
    g = ds.read_image(str(tmp_path / "grey.png"))                      # .convert('RGB'), vqa_dataset.py:38
    assert g.shape == (11, 13, 3) and np.array_equal(g[:, :, 1], a[:, :, 0])
    np.save(str(tmp_path / "bad.npy"), a.astype(np.float32))
    with pytest.raises(ValueError):
        ds.read_image(str(tmp_path / "bad.npy"))


# ------------------------------------------------------------------------------------------------------------- GPU
def _make_image_set(directory, n, seed=0, hw=(480, 640)):
    os.makedirs(directory, exist_ok=True)
    r = np.random.RandomState(seed)
    arrays = []
    for i in range(n):
        h, w = hw if i % 3 else (hw[1], hw[0] - 40 * (i % 5))          # portrait and landscape, several sizes
        a = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
        np.save(os.path.join(directory, "img{}.npy".format(i)), a)
        arrays.append(a)
    return arrays


@pytest.mark.gpu
def test_uint8_sources_deliver_the_oracles_tensors(tmp_path):
    """Files -> prefetch threads -> pinned upload -> csrc/image.hip == Pillow's resize + ToTensor + Normalize, bit for bit."""
    from oracle import pil_resize
    arrays = _make_image_set(str(tmp_path / "img"), 7)
    ann = [{"question_id": 100 + i, "image": "img/img{}.npy".format(i), "words": [[2054], [2003]]} for i in range(7)]
    qfile = str(tmp_path / "q.json")
    with open(qfile, "w") as fh:
        json.dump(ann, fh)
    src = ds.VqaFilePairs(qfile, str(tmp_path), "vlmo", 8, 32)
    src.prefetch([4, 1, 6])
    x = src.images([4, 1, 6], "cuda:0")
    for row, i in zip(x, [4, 1, 6]):
        want = pil_resize.to_tensor_normalize(pil_resize.resize_bicubic_u8(arrays[i], 32, 32))
        assert np.array_equal(row.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert src.seconds_images > 0.0
    src.close()
    syn = ds.SyntheticUint8Pairs(5, 8, 32, "vlmo", seed=3, source_hw=(48, 64))
    y = syn.images([0, 3], "cuda:0")
    want = pil_resize.to_tensor_normalize(pil_resize.resize_bicubic_u8(syn._load_one(3), 32, 32))
    assert np.array_equal(y[1].cpu().numpy().view(np.uint32), want.view(np.uint32))
    syn.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry_run(out_path, argv):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "entry"))
    out = open(out_path, "w")
    os.dup2(out.fileno(), 1)
    os.dup2(out.fileno(), 2)
    sys.argv = ["run.py"] + list(argv)
    import run
    run.main()
    sys.stdout.flush()


@pytest.mark.gpu
def test_entry_run_on_files_equals_the_attack_fed_the_oracles_tensors(tmp_path, text_meta):
    """64 uint8 480 x 640 (and other) images + a questions file + tables + a reference-format CHECKPOINT on disk ->
    ``entry/run.py`` writes ``<question_id>.pt`` and ``adv_txt.json``; the same sweep fed the Pillow-oracle tensors of the
    same files (no file pipeline, no prefetch) gives the identical adversarial images and text."""
    from oracle import pil_resize
    from tests.golden import encoder_cases as ec
    from vqattack_amd.attack.runner import AttackConfig
    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox import checkpoint as ck
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters
    n = 64
    arrays = _make_image_set(str(tmp_path / "val2014"), n, seed=5)
    vocab = _vocab_file(tmp_path, text_meta["vocab"])
    body = [w for w in text_meta["vocab"][104:] if w.isalpha()]
    r = np.random.RandomState(1)
    ann = []
    for i in range(n):
        words = [body[j] for j in r.randint(0, len(body), r.randint(3, 7))]
        ann.append({"question_id": 4000 + 3 * i, "image": "val2014/img{}.npy".format(i), "question": " ".join(words) + "?",
                    "dataset": "vqa"})
    qfile = str(tmp_path / "vqa_val.json")
    with open(qfile, "w") as fh:
        json.dump(ann, fh)
    keep = [a["question_id"] for a in ann if a["question_id"] % 5]         # right_part filters some questions out
    tables_dir = str(tmp_path / "tables")
    os.makedirs(tables_dir)
    with open(os.path.join(tables_dir, "right_part.txt"), "w") as fh:
        fh.write("".join("{}\n".format(q) for q in keep))
    # a reference-format checkpoint (the encoder fixture's tiny VLMo, 997-word vocabulary >= this test's 234)
    with open(os.path.join(GOLD, "encoder_golden.json")) as fh:
        rec = json.load(fh)["vlmo_tiny"]
    sd = ec.seeded_state_dict(rec["listing"], rec["seed"])
    ckpt = str(tmp_path / "vlmo_tiny_pretrain.pt")
    torch.save({"state_dict": sd}, ckpt)
    # the candidate proposer: a BertForMaskedLM state dict in the transformers library's own key names (adv_attack.py:110)
    tf = pytest.importorskip("transformers")
    torch.manual_seed(11)
    hf = tf.BertForMaskedLM(tf.BertConfig(vocab_size=len(text_meta["vocab"]), hidden_size=128, num_hidden_layers=2,
                                          num_attention_heads=2, intermediate_size=512, max_position_embeddings=64))
    with torch.no_grad():
        for name, prm in hf.named_parameters():
            if prm.dim() == 2 and "embeddings" not in name:
                prm.normal_(0.0, prm.shape[1] ** -0.5)
    mlm_ckpt = str(tmp_path / "bert_mlm.pt")
    torch.save(hf.state_dict(), mlm_ckpt)
    attack_dir = str(tmp_path / "attack_dir_VLMO_BASE")
    out = str(tmp_path / "run.out")
    argv = ["with", "image_size=32", "max_text_len=40", "per_gpu_batchsize=16", "mixed=True", "questions=" + qfile,
            "image_root=" + str(tmp_path), "vocab_file=" + vocab, "tables_dir=" + tables_dir, "pretrain_path=" + ckpt,
            "mlm_checkpoint=" + mlm_ckpt, "attack_dir=" + attack_dir, "seed=3", "sim_threshold=0.2"]
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=_entry_run, args=(out, argv))
    p.start()
    p.join(timeout=600)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("entry/run.py did not finish within 600 s")
    text = open(out).read()
    assert p.exitcode == 0, text[-3000:]
    acc = [ln for ln in text.splitlines() if ln.startswith("acc_vqa")]
    assert len(acc) == 1 and int(acc[0].split()[2]) == len(keep)
    adv_txt = json.load(open(os.path.join(attack_dir, "adv_txt.json")))
    assert sorted(map(int, adv_txt)) == sorted(keep)
    assert sorted(os.listdir(attack_dir)) == sorted(["adv_txt.json"] + ["{}.pt".format(q) for q in keep])

    # the same attack, fed the oracle's tensors: same models (same checkpoint, same seeds), same plan, no file pipeline
    dev = torch.device("cuda", 0)
    white = ck.vlmo_from_reference(torch.load(ckpt, weights_only=True), image_size=32, vqa_head=False).to(dev)
    black = FrozenVlmo.finetuned_from(white, seed=3 + 1).to(dev)
    src = ds.VqaFilePairs(qfile, str(tmp_path), "vlmo", 40, 32, tokenizer=WordPiece(vocab),
                          tables=ds.load_tables(tables_dir, "vlmo"))
    by_qid = {a["question_id"]: arrays[i] for i, a in enumerate(ann)}

    def oracle_images(indices, device):
        rows = [pil_resize.to_tensor_normalize(pil_resize.resize_bicubic_u8(by_qid[src.qids[i]], 32, 32)) for i in indices]
        return torch.from_numpy(np.stack(rows)).to(device)
    src.images, src.prefetch = oracle_images, (lambda indices: None)
    ref_dir = str(tmp_path / "ref_dir")
    torch.manual_seed(3)                             # entry/run.py seeds torch with seed + rank (VQA.py:74-77)
    from vqattack_amd.attack.proposer import BertMlmProposer, banned_ids
    proposer = BertMlmProposer.from_hf_state_dict(torch.load(mlm_ckpt, weights_only=True)).to(dev)
    banned = banned_ids(WordPiece(vocab).tokens, ds.DEFAULT_STOP_WORDS).to(dev)
    res = run_sweep("vlmo", white, black, VlmoAttackAdapters(white), 0, 16, 32, 40, dev, mixed=True, save_dir=ref_dir,
                    log_every=0, source=src, config=AttackConfig(sim_threshold=0.2), mlm_logits_fn=proposer,
                    banned_ids=banned)
    assert any(row != src.ids[i].tolist() for i, row in enumerate(res["adv_text"][str(q)] for q in src.qids)), \
        "the proposer's candidates never led to a substitution: the text side of the test would be vacuous"
    assert res["n_total"] == len(keep) and res["skipped"] == n - len(keep)
    assert {k: v for k, v in res["adv_text"].items()} == adv_txt
    for q in keep:
        a = torch.load(os.path.join(attack_dir, "{}.pt".format(q)))
        b = torch.load(os.path.join(ref_dir, "{}.pt".format(q)))
        assert a.shape == (1, 3, 32, 32) and torch.equal(a, b), q


def _entry_vqa(out_path, argv):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "entry"))
    out = open(out_path, "w")
    os.dup2(out.fileno(), 1)
    os.dup2(out.fileno(), 2)
    sys.argv = ["VQA.py"] + list(argv)
    import VQA
    VQA.main()
    sys.stdout.flush()


@pytest.mark.gpu
def test_entry_vqa_on_files_with_a_reference_format_albef_checkpoint(tmp_path, text_meta):
    """The ALBEF-flavor entry point (argparse + yaml, ALBEF_attack/VQA.py:119-134) on files: annotation json, images,
    vocabulary, ``right_part`` filter and a pre-trained checkpoint in the reference's key layout (``visual_encoder.*``,
    ``text_encoder.bert.*``, ``text_encoder.cls.*``; adv_attack.py:83-92) -> ``attack_dir/<qid>.pt`` + ``adv_txt.json``,
    identical to the same sweep fed the Pillow oracle's tensors (the white box re-draws its 15 % token mask in every
    forward, model_pretrain.py:130-132: both runs start from the seed the entry point sets)."""
    import yaml
    from oracle import pil_resize
    from tests.golden import encoder_cases as ec
    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox import checkpoint as ck
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef
    n = 24
    arrays = _make_image_set(str(tmp_path / "val2014"), n, seed=9, hw=(240, 320))
    vocab = _vocab_file(tmp_path, text_meta["vocab"])
    body = [w for w in text_meta["vocab"][104:] if w.isalpha()]
    r = np.random.RandomState(2)
    ann = [{"question_id": 700 + 11 * i, "image": "val2014/img{}.npy".format(i), "dataset": "vqa",
            "question": " ".join(body[j] for j in r.randint(0, len(body), r.randint(3, 7))) + "?"} for i in range(n)]
    qfile = str(tmp_path / "vqa_val.json")
    with open(qfile, "w") as fh:
        json.dump(ann, fh)
    keep = [a["question_id"] for a in ann if a["question_id"] % 4]
    tables_dir = str(tmp_path / "tables")
    os.makedirs(tables_dir)
    with open(os.path.join(tables_dir, "right_part.txt"), "w") as fh:
        fh.write("".join("{}\n".format(q) for q in keep))
    with open(os.path.join(GOLD, "encoder_golden.json")) as fh:
        rec = json.load(fh)["albef_tiny"]
    tied = set(rec["tied"])
    sd = ec.seeded_state_dict([e for e in rec["listing"] if e[0] not in tied], rec["seed"])
    ckpt = str(tmp_path / "ALBEF_pretrain.pth")
    torch.save({"model": sd}, ckpt)
    cfg_path = str(tmp_path / "VQA.yaml")
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(dict(image_res=32, batch_size_test=8, text_len=12, n_samples=0, attack_dir="attack_dir",
                            vqa_root=str(tmp_path)), fh)
    out_dir = str(tmp_path / "out")
    out = str(tmp_path / "vqa.out")
    argv = ["--config", cfg_path, "--output_dir", out_dir, "--seed", "5", "--mixed", "--questions", qfile, "--vocab_file",
            vocab, "--tables_dir", tables_dir, "--checkpoint", ckpt, "--sim_threshold", "0.2"]
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=_entry_vqa, args=(out, argv))
    p.start()
    p.join(timeout=600)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("entry/VQA.py did not finish within 600 s")
    text = open(out).read()
    assert p.exitcode == 0, text[-3000:]
    adv_txt = json.load(open(os.path.join(out_dir, "adv_txt.json")))
    assert sorted(map(int, adv_txt)) == sorted(keep)
    attack_dir = os.path.join(out_dir, "attack_dir")
    assert sorted(os.listdir(attack_dir)) == sorted("{}.pt".format(q) for q in keep)
    # the same sweep, fed the oracle's tensors
    dev = torch.device("cuda", 0)
    white = ck.albef_from_reference(torch.load(ckpt, weights_only=True), image_size=32, vqa_head=False).to(dev)
    assert white.cfg.heads == 1                       # the entry point cannot know the head count of a 64-wide test model
    black = FrozenAlbef.finetuned_from(white, seed=5 + 1).to(dev)
    src = ds.VqaFilePairs(qfile, str(tmp_path), "albef", 12, 32, tokenizer=WordPiece(vocab),
                          tables=ds.load_tables(tables_dir, "albef"))
    by_qid = {a["question_id"]: arrays[i] for i, a in enumerate(ann)}

    def oracle_images(indices, device):
        rows = [pil_resize.to_tensor_normalize(pil_resize.resize_bicubic_u8(by_qid[src.qids[i]], 32, 32)) for i in indices]
        return torch.from_numpy(np.ascontiguousarray(np.stack(rows))).to(device)
    src.images, src.prefetch = oracle_images, (lambda indices: None)
    ref_dir = str(tmp_path / "ref_dir")
    torch.manual_seed(5)
    from vqattack_amd.attack.runner import AttackConfig
    res = run_sweep("albef", white, black, AlbefAttackAdapters(white), 0, 8, 32, 12, dev, mixed=True, save_dir=ref_dir,
                    log_every=0, source=src, config=AttackConfig(sim_threshold=0.2))
    assert res["adv_text"] == adv_txt
    for q in keep:
        a = torch.load(os.path.join(attack_dir, "{}.pt".format(q)))
        b = torch.load(os.path.join(ref_dir, "{}.pt".format(q)))
        assert torch.equal(a, b), q
