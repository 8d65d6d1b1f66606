"""MI355X-native SDT voice2pose training hot path (HIP kernels behind the reference core.networks / core.pipelines API)."""
