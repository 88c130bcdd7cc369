"""Loss-weight annealer with the reference's interface (utils/weight_scheduler.py:4-39): named weights
{name: {value, anneal_rate[, min][, max]}}, multiplied by their rate on step(), clamped, readable as attributes.
The reference steps it every epoch but never uses the values in a loss expression; kept for config compatibility."""
import copy


class WeightScheduler:
    def __init__(self, weight_groups):
        object.__setattr__(self, "_initial", copy.deepcopy(dict(weight_groups)))
        object.__setattr__(self, "weights", copy.deepcopy(dict(weight_groups)))

    def reset(self):
        object.__setattr__(self, "weights", copy.deepcopy(self._initial))

    def step(self):
        for info in self.weights.values():
            v = info['value'] * info['anneal_rate']
            if 'min' in info:
                v = max(v, info['min'])
            if 'max' in info:
                v = min(v, info['max'])
            info['value'] = v

    def __getattr__(self, name):
        try:
            return self.weights[name]['value']
        except KeyError:
            raise AttributeError(name)
