"""MI355X-native DQ-VAE / DQ-Transformer hot path."""
