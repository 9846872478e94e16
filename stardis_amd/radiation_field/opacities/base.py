"""`Opacities` container, call-compatible with stardis/radiation_field/opacities/base.py:4-28.

The dictionary holds host (numpy) arrays exactly like the reference's; the arrays our own
solvers produce also stay resident on the GPU, and the total is summed there in the
reference's order (dictionary insertion order, `+=` per entry, entries named *gammas* /
*doppler* skipped)."""
import numpy as np

from stardis_amd import ops
from stardis_amd._lib import default_context


class Opacities:
    def __init__(self, frequencies, stellar_model):
        self.opacities_dict = {}
        self.total_alphas = np.zeros((stellar_model.no_of_depth_points, len(frequencies)))
        self._resident = {}  # key -> (host array object, device array) for entries computed by this package
        self._total_dev = None

    def _remember(self, key, host_array, device_array):
        self._resident[key] = (host_array, device_array)

    def _device_entry(self, ctx, key, value):
        cached = self._resident.get(key)
        if cached is not None and cached[0] is value:
            return cached[1]
        return ctx.upload(np.broadcast_to(np.asarray(value, dtype=np.float64), self.total_alphas.shape))

    def calc_total_alphas(self):
        ctx = default_context()
        # += into total_alphas (idempotence is NOT guaranteed, like the reference); the usual all-zero start needs no upload
        total = ctx.upload(self.total_alphas) if np.any(self.total_alphas) else ctx.zeros(self.total_alphas.shape)
        for key, value in self.opacities_dict.items():
            if "gammas" in key or "doppler" in key:
                continue
            if np.ndim(value) == 0 and value == 0:
                continue  # a disabled source returns the scalar 0 (base.py:164-165, :359-360)
            ops.accumulate(total, self._device_entry(ctx, key, value), ctx)
        self.total_alphas[...] = total.numpy()
        self._total_dev = (self.total_alphas.copy(), total)
        return self.total_alphas

    def total_alphas_device(self, ctx):
        """Device copy of total_alphas, reused if the host array is unchanged since calc_total_alphas."""
        if self._total_dev is not None and np.array_equal(self._total_dev[0], self.total_alphas):
            return self._total_dev[1]
        return ctx.upload(self.total_alphas)
