from .AgentBase import AgentBase
from .AgentPPO import AgentPPO, AgentDiscretePPO, ActorPPO, ActorDiscretePPO, CriticPPO
from .AgentSAC import AgentSAC, ActorSAC, CriticEnsemble
