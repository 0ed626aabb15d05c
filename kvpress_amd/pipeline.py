"""KVPressTextGenerationPipeline: the caller of the hot path.

Same user-facing behaviour as the reference's kvpress/pipeline.py (KVPressTextGenerationPipeline :25-323,
registered as "kv-press-text-generation" :327-331): the context is pre-filled ONCE through ``model.model`` (no
lm_head) under ``with press(model):`` -- every attention layer's K/V is pruned by the press's HIP kernels as soon as
the layer has run -- and each question is then answered by greedy decoding on top of the compressed cache, which is
cut back to its post-prefill length between questions.

    pipe = pipeline("kv-press-text-generation", model=model, tokenizer=tokenizer)
    pipe(context, question="...", press=SnapKVPress(0.5))["answer"]
    pipe(context, questions=[...], press=...)["answers"]

Host-side differences to the reference, none of which changes a result:
  * FinchPress (delimiter token, :224-232), DecodingPress, PrefillDecodingPress and KeyRerotationPress are handled as there; the
    DMSPress / RestoreKVPress special cases (:230-232, :243) are absent (those presses are not part of this package);
  * ``logits_to_keep`` is the transformers >= 4.50 name of ``num_logits_to_keep`` (:288).
"""
from __future__ import annotations

import contextlib
import logging
from typing import Optional

import torch
from transformers import AutoModelForCausalLM, Cache, DynamicCache, Pipeline
from transformers.pipelines import PIPELINE_REGISTRY

from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.decoding_press import DecodingPress, PrefillDecodingPress
from kvpress_amd.presses.finch_press import FinchPress
from kvpress_amd.presses.key_rerotation_press import KeyRerotationPress

logger = logging.getLogger(__name__)

TASK = "kv-press-text-generation"


class KVPressTextGenerationPipeline(Pipeline):
    """Compress the KV cache of a long context while pre-filling, then answer questions about it greedily."""

    # ---- transformers.Pipeline plumbing -------------------------------------------------------
    def _sanitize_parameters(self, question: Optional[str] = None, questions: Optional[list[str]] = None,
                             answer_prefix: Optional[str] = None, press: Optional[BasePress] = None,
                             max_new_tokens: int = 50, max_context_length: Optional[int] = None,
                             enable_thinking: bool = False, cache: Optional[Cache] = None, **kwargs):
        """Split the call's keyword arguments over preprocess / _forward / postprocess (pipeline.py:40-107).

        question / questions are exclusive; with neither, one empty question is asked (the answer then continues
        the context)."""
        assert question is None or questions is None, "Either question or questions should be provided, not both."
        single = questions is None
        if questions is None:
            questions = [question] if question else [""]
        if max_context_length is None:
            max_context_length = min(self.tokenizer.model_max_length, int(1e10))
        pre = dict(questions=questions, answer_prefix=answer_prefix or "", max_context_length=max_context_length,
                   enable_thinking=enable_thinking)
        fwd = dict(press=press, max_new_tokens=max_new_tokens, cache=cache)
        post = dict(single_question=single)
        return pre, fwd, post

    def preprocess(self, context: str, questions: list[str], answer_prefix: str, max_context_length: int,
                   enable_thinking: bool = False):
        """Chat-template and tokenise context and questions separately (pipeline.py:109-171): the context ends where
        the user turn's content would end, each question carries the rest of the template (end of the user turn +
        generation prompt) and the answer prefix."""
        tok = self.tokenizer
        if tok.chat_template is None:
            context = (getattr(tok, "bos_token", "") or "") + context
            suffix = "\n"  # separates the question from the answer
        else:
            marker = "#" * (len(context) + 10)  # cannot occur in the context
            rendered = tok.apply_chat_template([{"role": "user", "content": context + marker}], add_generation_prompt=True,
                                               tokenize=False, enable_thinking=enable_thinking)
            context, suffix = rendered.split(marker)

        def encode(text):
            return tok.encode(text, return_tensors="pt", add_special_tokens=False)

        context_ids = encode(context)
        if context_ids.shape[1] > max_context_length:
            logger.warning(f"Context length has been truncated from {context_ids.shape[1]} to {max_context_length} tokens.")
            context_ids = context_ids[:, :max_context_length]
        return {"context_ids": context_ids, "questions_ids": [encode(q + suffix + answer_prefix) for q in questions]}

    def postprocess(self, model_outputs, single_question):
        return {"answer": model_outputs[0]} if single_question else {"answers": model_outputs}

    # ---- the work -----------------------------------------------------------------------------
    def _forward(self, input_tensors, max_new_tokens: int = 50, press: Optional[BasePress] = None,
                 cache: Optional[Cache] = None):
        """Prefill the context under the press, then one greedy answer per question (pipeline.py:173-246)."""
        is_decoding_press = isinstance(press, (DecodingPress, PrefillDecodingPress))
        decoding = is_decoding_press   # pipeline.py:230-232 (the reference also counts a decoding DMSPress: not part of this package)
        if is_decoding_press and len(input_tensors["questions_ids"]) > 1:
            raise ValueError("DecodingPress is not compatible with multiple questions. Please specify a single question.")
        device = self.model.device
        context_ids = input_tensors["context_ids"].to(device)
        context_length = context_ids.shape[1]
        if cache is None:
            cache = DynamicCache()

        # a pure DecodingPress does nothing during prefill (pipeline.py:217-219)
        prefill_press = press if press is not None and not isinstance(press, DecodingPress) else None
        with prefill_press(self.model) if prefill_press is not None else contextlib.nullcontext():
            self.model.model(input_ids=context_ids, past_key_values=cache)  # no lm_head during prefill
        logger.debug(f"Context Length: {context_length}")
        logger.debug(f"Compressed Context Length: {cache.get_seq_length()}")

        answers = []
        # decoding presses keep their hook for the answers (pipeline.py:230-233)
        with press(self.model) if decoding else contextlib.nullcontext():
            for question_ids in input_tensors["questions_ids"]:
                if isinstance(press, KeyRerotationPress) or (isinstance(press, FinchPress) and press.rerotate_keys):
                    context_length = cache.get_seq_length()  # re-rotated keys sit at positions 0..n-1 (:237-238)
                kept = [cache.get_seq_length(i) for i in range(len(cache))]
                answers.append(self.generate_answer(question_ids.to(device), cache, context_length, max_new_tokens))
                self._remove_answer_from_cache(cache, kept)
        return answers

    @staticmethod
    def _remove_answer_from_cache(cache: Cache, lengths: list[int]):
        """Cut every layer back to the length it had before the question (pipeline.py:248-263); layers may differ in
        length (per-layer budgets)."""
        for layer, n in zip(cache.layers, lengths):
            layer.keys = layer.keys[:, :, :n]
            layer.values = layer.values[:, :, :n]
            if hasattr(layer, "_quantized_keys"):
                layer._quantized_keys = layer._quantized_keys[:, :, :n]
                layer._quantized_values = layer._quantized_values[:, :, :n]

    def generate_answer(self, question_ids: torch.Tensor, cache: Cache, context_length: int, max_new_tokens: int) -> str:
        """Greedy decoding of one answer (pipeline.py:265-317).  Positions continue from the ORIGINAL context length:
        the pruned cache keeps each key's own rotary phase, so new tokens must sit after the uncompressed context."""
        device = self.model.device
        pos = torch.arange(context_length, context_length + question_ids.shape[1], device=device).unsqueeze(0)
        out = self.model(input_ids=question_ids, past_key_values=cache, position_ids=pos, logits_to_keep=1)
        token = out.logits[0, -1].argmax()
        generated = [token]

        stop = self.model.generation_config.eos_token_id
        stop = [] if stop is None else (stop if isinstance(stop, (list, tuple)) else [stop])
        stop_ids = torch.tensor(stop, device=device, dtype=token.dtype) if stop else None
        next_pos = pos[:, -1:] + 1
        # As in the reference, the stop test applies to the tokens produced inside this loop (not to the first one), the
        # stop token itself is kept, and no forward pass runs after it (a decoding press would otherwise see extra steps).
        for i in range(max_new_tokens - 1):
            out = self.model(input_ids=token.view(1, 1), past_key_values=cache, position_ids=next_pos)
            token = out.logits[0, -1].argmax()
            generated.append(token)
            next_pos = next_pos + 1
            if stop_ids is not None and bool(torch.isin(token, stop_ids)):
                break
        return str(self.tokenizer.decode(torch.stack(generated), skip_special_tokens=True))


PIPELINE_REGISTRY.register_pipeline(TASK, pipeline_class=KVPressTextGenerationPipeline, pt_model=AutoModelForCausalLM)
