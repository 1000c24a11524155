"""The streaming detokenizer of tiny_llm_hip.loader on a BYTE-LEVEL BPE tokenizer (Qwen's kind; the fixture tokenizer of the other
loader tests is word-level): a multi-byte UTF-8 character can be split over several tokens, and `mlx_lm`'s detokenizers -- whose
surface the generation loops use (reference generate.py / batch.py: reset, add_token, last_segment, text, finalize) -- hold such a
character back until its last byte arrives.  Built with the `tokenizers` library here (few merges, so that most characters ARE
split): the segments emitted token by token concatenate to the full decode and never contain U+FFFD."""
import pytest

tokenizers = pytest.importorskip("tokenizers")
transformers = pytest.importorskip("transformers")


def _byte_level_tokenizer():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=300, special_tokens=["<|endoftext|>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  show_progress=False)
    tok.train_from_iterator(["hello world tiny llm on mi355x", "the quick brown fox", "hello hello world"], trainer)
    return transformers.PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>")


@pytest.mark.parametrize("text", ["hello world", "naïve café — déjà vu", "日本語のテキスト と 中文", "emoji 🙂🚀 mixed with text 👍🏽!", "a\n\nb\tc  d"])
def test_segments_concatenate_to_the_decode_and_hold_back_partial_characters(text):
    from tiny_llm_hip.loader import TokenizerWrapper

    wrapper = TokenizerWrapper(_byte_level_tokenizer())
    ids = wrapper.encode(text)
    assert wrapper.decode(ids) == text
    if any(ord(ch) > 127 for ch in text):
        assert len(ids) > len(text.encode("utf-8")) // 3, "the tokenizer was meant to split multi-byte characters"
    detok = wrapper.detokenizer
    detok.reset()
    pieces = []
    for t in ids:
        detok.add_token(t)
        seg = detok.last_segment
        assert "�" not in seg, "half a character was emitted"
        pieces.append(seg)
    detok.finalize()
    assert "".join(pieces) == text == detok.text
    # a fresh detokenizer of the same class over the raw tokenizer, as the reference's batch loop builds one per request (batch.py:23)
    other = detok.__class__(wrapper._tokenizer)
    for t in ids[:3]:
        other.add_token(t)
    assert other.text == wrapper.decode(ids[:3])
