from .AgentBase import AgentBase
from .AgentPPO import AgentPPO, AgentA2C, AgentDiscretePPO, ActorPPO, ActorDiscretePPO, CriticPPO
from .AgentSAC import AgentSAC, ActorSAC, CriticEnsemble
