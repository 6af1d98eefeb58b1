from .AgentBase import AgentBase
from .AgentPPO import AgentPPO, AgentA2C, AgentDiscretePPO, AgentDiscreteA2C, ActorPPO, ActorDiscretePPO, CriticPPO
from .AgentSAC import AgentSAC, AgentModSAC, ActorSAC, ActorFixSAC, CriticEnsemble
