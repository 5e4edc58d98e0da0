"""Kernel-selection rules of the op layer that are plain functions of the shape (no GPU needed)."""
from llm_consensus_amd import ops
from llm_consensus_amd.models.config import get_preset


def gate_up_cols(name):
    return 2 * get_preset(name).intermediate_size


def test_wide_panel_workgroups_only_where_they_save_a_round():
    """160-column workgroups are for matrices of 257-320 column tiles of 128 whose K is not split:
    among the served models that is Qwen2.5-7B's gate/up projection (296 tiles -> 237 workgroups).
    Narrower matrices fit one round anyway, much wider ones need several rounds either way, and
    the rule stops at the panel kernel's 256 rows."""
    q, l = get_preset("qwen2.5-7b"), get_preset("llama-3-8b")
    assert gate_up_cols("qwen2.5-7b") // 128 == 296
    for rows in (1, 32, 64, 128, 256):
        assert ops.use_wide_panel(rows, gate_up_cols("qwen2.5-7b"), q.hidden_size)
        assert not ops.use_wide_panel(rows, gate_up_cols("llama-3-8b"), l.hidden_size)       # 224 tiles
        assert not ops.use_wide_panel(rows, l.vocab_size, l.hidden_size)                     # 1002 tiles
        assert not ops.use_wide_panel(rows, q.vocab_size, q.hidden_size)                     # 1188 tiles
        assert not ops.use_wide_panel(rows, 6144, 4096)                                      # QKV: K is split
    assert not ops.use_wide_panel(300, gate_up_cols("qwen2.5-7b"), q.hidden_size)
    assert ops.use_wide_panel(64, 320 * 128, 4096) and not ops.use_wide_panel(64, 321 * 128, 4096)


def test_four_wave_prefill_gemm_for_long_reductions_only():
    """The four-wave 256 x 256 kernel wins where its longer prologue is spread over many K-tiles
    (measured: K = 14336 +3 %, K = 4096 -1..-4 %): the down projections take it, the projections
    whose reduction runs over d_model do not, and fp32 outputs (split-K slabs) never do."""
    for name in ("llama-3-8b", "qwen2.5-7b", "mistral-7b", "llama-3-70b"):
        c = get_preset(name)
        assert ops.use_four_wave_gemm(c.intermediate_size)          # down: K = intermediate size
        assert not ops.use_four_wave_gemm(c.hidden_size) or c.hidden_size >= 8192   # QKV / O / gate-up
        assert not ops.use_four_wave_gemm(c.intermediate_size, out_f32=True)
    assert not ops.use_four_wave_gemm(4096) and ops.use_four_wave_gemm(8192)
