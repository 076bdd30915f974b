"""cover_ref — CPU restatement (plain PyTorch-CPU / numpy) of the reference's candidate-sampling-and-verification
path. TEST INFRASTRUCTURE ONLY: it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under cover_vla_amd/ imports it, and it is never the thing measured or shipped.

Parity pinning: the reference's own tests hold no golden vectors for this path, so
  * P1 (pi0 sampler + CoVer verifier): pinned against outputs of the reference's modules imported in the build
    container (oracle/gen_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py);
  * P2 (OpenVLA-7B shapes): no reference code exists -> "parity unpinned at the reference"; pinned instead against
    HF transformers' LlamaForCausalLM / Dinov2 / Siglip modules (oracle/gen_golden.py, same test file).
"""
