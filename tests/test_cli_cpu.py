"""CPU: the two entry scripts parse the REFERENCE's command lines (R/test_vllm_edit.py:7-18, R/train_vllm_editor.py:14-29) with no
extra required flag, and the optional path / encoder flags resolve as devqa_amd/cli.py documents."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_reference_command_lines_parse():
    import test_vllm_edit as T
    import train_vllm_editor as TR
    # README / gen_bash.py style invocations of the reference
    c = T.get_attr("-en ft_vl -mn blip2 -sen 1 -dvc cuda:0 -dn EVQA".split())
    assert (c.editor_name, c.edit_model_name, c.sequential_edit_n, c.device, c.data_name) == ("ft_vl", "blip2", 1, "cuda:0", "EVQA")
    assert c.data_sample_n is None and c.eval_name_postfix == "" and c.editor_ckpt_path is None
    c = T.get_attr("-en mend_vl -mn minigpt4 -sen 1000 -dvc cuda:0 -ckpt records/x/Best -dn VLKEB -dsn 500 -enp run1".split())
    assert (c.sequential_edit_n, c.editor_ckpt_path, c.data_sample_n, c.eval_name_postfix) == (1000, "records/x/Best", 500, "run1")
    t = TR.get_attr("-en mend_vl -mn blip2 -dna EVQA -bs 4 -dvc cuda:0".split())
    assert (t.batch_size, t.epochs, t.save_ckpt_per_i, t.log_per_i, t.ema_alpha, t.data_buffer_size, t.extra_devices) == \
        (4, 1000, 1000, 1, 0.1, 4, [0])
    t = TR.get_attr("-en mend_vl -mn blip2 -dna VLKEB -bs 2 -dvc cuda:0 -lkpt None -edvc 1 -eps 3 -tnp tag -rs 7 -dn 100".split())
    assert t.load_ckpt_path is None and t.extra_devices == [1] and t.random_seed == 7 and t.data_n == 100
    with pytest.raises(SystemExit):
        TR.get_attr("-en mend_vl".split())          # -mn -dna -bs -dvc are required there, as in the reference


def test_path_defaults_and_encoders(tmp_path, monkeypatch):
    import devqa_amd  # noqa: F401
    import test_vllm_edit as T
    from devqa_amd import cli
    monkeypatch.setenv("DEVQA_DATA_ROOT", "/d")
    monkeypatch.delenv("DEVQA_IMG_ROOT", raising=False)
    c = T.get_attr("-en ft_vl -mn blip2 -sen 1 -dvc cuda:0 -dn EVQA".split())
    assert cli.resolve_paths(c, "eval") == ("/d/easy-edit-mm/vqa/vqa_eval.json", "/d/easy-edit-mm/images", "/d/embeddings/vqa_embeddings.npz")
    assert cli.resolve_paths(c, "train")[0] == "/d/easy-edit-mm/vqa/vqa_train.json"
    c = T.get_attr("-en ft_vl -mn blip2 -sen 1 -dvc cuda:0 -dn vlkeb --img_root /imgs".split())
    assert cli.resolve_paths(c, "eval") == ("/d/VLKEB/eval.json", "/imgs", "/d/embeddings/vlkeb_embeddings.npz")
    c.data_name = "EIC"; c.img_root = None                                               # R/test_vllm_edit.py:50-54
    assert cli.resolve_paths(c, "eval") == ("/d/easy-edit-mm/caption/caption_eval_edit.json", "/d/easy-edit-mm/images",
                                            "/d/embeddings/caption_embeddings.npz")
    c.data_name = "COCO"
    with pytest.raises(BaseException):
        cli.resolve_paths(c, "eval")
    # encoders: none -> a clear error from the dataset builder; lookup table; module:callable
    c = T.get_attr("-en ft_vl -mn blip2 -sen 1 -dvc cuda:0 -dn EVQA".split())
    assert cli.load_encoder(c) is None
    with pytest.raises(BaseException, match="sentence encoder"):
        cli.build_dataset(c, "eval")
    np.savez(tmp_path / "q.npz", sentences=np.asarray(["a b", "c"]), embeddings=np.eye(2, 4, dtype=np.float32))
    c.queries = str(tmp_path / "q.npz")
    enc = cli.load_encoder(c)
    assert enc(["c", "a b"]).tolist() == [[0, 1, 0, 0], [1, 0, 0, 0]]
    with pytest.raises(BaseException, match="no pre-computed embedding"):
        enc(["zzz"])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    c.encoder = "retr_common:encode"
    assert cli.load_encoder(c)(["x y"]).shape == (1, 96)
    # editor extras: IKE needs corpus sentences + encoder, TP its text file, FT none
    assert cli.editor_kwargs(c) == {}
    c.editor_name = "ike_vl"
    np.savez(tmp_path / "ike.npz", sentences=np.asarray(["s1", "s2"]), embeddings=np.zeros((2, 96), np.float32))
    c.ike_corpus = str(tmp_path / "ike.npz")
    kw = cli.editor_kwargs(c)
    assert kw["corpus"]["sentences"] == ["s1", "s2"] and callable(kw["encode"])
    c.editor_name = "tp_vl"
    with pytest.raises(BaseException, match="tp_texts"):
        cli.editor_kwargs(c)
