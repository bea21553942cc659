"""CPU: IKE corpus / prompt construction (strings only; retrieval arithmetic is tested on the GPU)."""
import numpy as np

import devqa_amd  # noqa: F401
from devqa_amd.editor.vllm_editors.ike_vl.ike_vl import build_ike_corpus, ike_sentence


def test_corpus_layout_and_sentence_format():
    recs = [{"prompt": "What color is the bus?", "target": "red", "rephrase_prompt": "The bus has which color?",
             "locality_prompt": "nq question: who wrote hamlet", "locality_ground_truth": "Shakespeare",
             "image_path": "a.jpg", "rephrase_image_path": "b.png", "locality_image_path": "c.jpg"}]
    c = build_ike_corpus(recs, lambda s: np.zeros((len(s), 4), np.float32))
    nf = "What color is the bus? red"
    assert c["sentences"] == [
        "New Fact: %s\nPrompt: %s\n\n" % (nf, nf),
        "New Fact: %s\nPrompt: The bus has which color? red\n\n" % nf,
        "New Fact: %s\nPrompt: nq question: who wrote hamlet Shakespeare\n\n" % nf]
    assert c["images"] == ["a.jpg", "b.png", "c.jpg"] and c["prompts"][2] == ["nq question: who wrote hamlet", "Shakespeare"]
    assert c["embeddings"].shape == (3, 4)
    assert ike_sentence("f", "p") == "New Fact: f\nPrompt: p\n\n"
