from .AgentBase import AgentBase
from .AgentPPO import AgentPPO, ActorPPO, CriticPPO
from .AgentSAC import AgentSAC, ActorSAC, CriticEnsemble
