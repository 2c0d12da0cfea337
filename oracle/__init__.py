"""CPU oracle for the VGGT -> Perceiver -> Qwen3 hot path.

TEST INFRASTRUCTURE ONLY. Nothing under vggt_qwen3_amd/ (the product) imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker / reported baseline.

It is a plain PyTorch-CPU restatement (torch functional ops on state-dict tensors, no nn.Module reuse) of the
reference's algorithm:
  qwen3.py      transformers/models/qwen3/modeling_qwen3.py:49-64,81-83,104-170,185-207,241-280,294-323,367-427,448-508
                + transformers/loss/loss_utils.py:32-71
  perceiver.py  /root/reference/src/models/projector_perceiver.py:30-82 (+ torch MHA math path,
                torch/nn/functional.py `multi_head_attention_forward`)
  vlm.py        /root/reference/src/models/vggt_qwen3_vlm.py:128-201 (encode_images slice, encode_geom, splice, loss)
  vggt.py       the un-vendored `vggt` package's Aggregator, restated from its published architecture
  collate.py    /root/reference/src/dataio/collate_multiview.py:46-79 (token ids / labels / mask layout)
  preprocess.py Pillow's two-pass bicubic resampler + torchvision's Resize/CenterCrop integer rules (collate_multiview.py:12-19)
  generate.py   transformers' greedy generate() with RepetitionPenalty / NoRepeatNGram processors (qa_inference.py:207-216)
  fp8.py        the e4m3 forward contract of BASELINE config C5 (no reference code exists: parity unpinned)

Pinning: qwen3/perceiver/vlm are pinned by tests/golden/*.npz, generated in the build container by
tools/make_golden.py from the reference's own modules (imported from /root/reference) and HF transformers'
Qwen3ForCausalLM; collate by the reference collator's ids; preprocess by Pillow itself; generate by transformers' own
generate() (tools/make_golden_generate.py). vggt.py: the DINOv2-with-registers stage is pinned against transformers'
Dinov2WithRegistersModel (tools/make_golden_dinov2.py); the alternating frame/global stage is PARITY UNPINNED - the
reference does not vendor the package or any test vector for it.
"""
