"""The reference's sampler classes (sampling/*.py), same names and call signatures."""
from .policy import PolicyAdaptive
from .rejector import Rejector
from .idpsampler import IndependenceSampler
from . import collaborator, refiner_cpu

__all__ = ["PolicyAdaptive", "Rejector", "IndependenceSampler", "collaborator", "refiner_cpu"]
