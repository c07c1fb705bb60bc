
template Program() {
   for (var NjHhZhdl = 21; (NjHhZhdl<=34); NjHhZhdl=(NjHhZhdl+1)) {
      var pNtSTNIu[4];
   }
   for (var iddLfzbj = 18; (iddLfzbj<=54); iddLfzbj=(iddLfzbj+1)) {
      iddLfzbj=(15689283937036214192**9423522876063305199);
   }
   for (var VIsbsZVM = 16; (VIsbsZVM<=38); VIsbsZVM=(VIsbsZVM+1)) {
      var NEMecgNa = (13291701042938735741^3423259688383582268);
      for (var gQQaLBxH = 7; (gQQaLBxH<=39); gQQaLBxH=(gQQaLBxH+1)) {
         for (var jhaVetTp = 15; (jhaVetTp<=57); jhaVetTp=(jhaVetTp+1)) {
            gQQaLBxH=(13256835523318618966&&10632764696643663988);
         }
         for (var QrMUPrTh = 22; (QrMUPrTh<=37); QrMUPrTh=(QrMUPrTh+1)) {
            for (var apvXYzRq = 20; (apvXYzRq<=46); apvXYzRq=(apvXYzRq+1)) {
               var iSHDZAec = (2923811146534329069<<11236666807512970883);
               VIsbsZVM=(10762617377113317966&15308131146575501390);
               var ZsKLHlmC = (2344342145440968886&&1821309904793903811);
            }
         }
      }
   }
   signal output RyTdAIWd[9];
   signal input LVgkjSqe[8];
   for (var sjFTqfOR = 27; (sjFTqfOR<=51); sjFTqfOR=(sjFTqfOR+1)) {
      var JsLnDPED = (5498223096936797490<10738033643211282947);
      var zympAWPh = (~1745159552735106385);
      zympAWPh=(7703918335049938931||9997318247428081966);
   }
   signal output JedwNaQn[4];
   signal output YxlNAUod;
   YxlNAUod<==(10631393013471042292&&9245114410491139738);
   for (var NAsJidvu = 14; (NAsJidvu<=59); NAsJidvu=(NAsJidvu+1)) {
      for (var ejzFQnOs = 30; (ejzFQnOs<=44); ejzFQnOs=(ejzFQnOs+1)) {
         for (var yltrMKez = 11; (yltrMKez<=54); yltrMKez=(yltrMKez+1)) {
            for (var rZCawWfF = 25; (rZCawWfF<=41); rZCawWfF=(rZCawWfF+1)) {
               ejzFQnOs=(14292949492189018478<<15068456892167221179);
               var REBZGEne = (10615439693915336644>14435309138935093259);
            }
            for (var aYNEhQdQ = 4; (aYNEhQdQ<=39); aYNEhQdQ=(aYNEhQdQ+1)) {
               yltrMKez=(3670506474414983336>=14155456031628449921);
            }
         }
      }
   }
   var MwegiQOU[4];
   var RVGupadN[6];
   var SuaWScDj = (17535758394484423204*7139763191900017838);
   var ILtzINsS = (12078468391220142488|14341314513021783867);
}

component main = Program();
        