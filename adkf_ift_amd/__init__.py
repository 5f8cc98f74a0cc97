"""adkf_ift_amd - MI355X-native ADKF-IFT inner-loop hot path (GNN deep-kernel features -> exact-GP marginal
likelihood -> IFT hypergradient) behind the reference's operator surface.  See DESIGN.md."""
__all__ = ["gp_ops", "hypergradient", "models", "stateless", "synthetic", "trainer", "roofline"]
