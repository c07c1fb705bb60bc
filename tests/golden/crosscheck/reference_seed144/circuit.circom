
template Program() {
   for (var kwVxXLHX = 23; (kwVxXLHX<=32); kwVxXLHX=(kwVxXLHX+1)) {
      signal input HZUrFfsj[3];
      kwVxXLHX=(1749199907588531778/8355700343131395676);
      signal output iTqvaMmo[2];
   }
   var CpUTWTUG[1];
   for (var DALQPAKb = 13; (DALQPAKb<=51); DALQPAKb=(DALQPAKb+1)) {
      for (var EQyarcMK = 30; (EQyarcMK<=48); EQyarcMK=(EQyarcMK+1)) {
         for (var OOCATveP = 20; (OOCATveP<=54); OOCATveP=(OOCATveP+1)) {
            DALQPAKb=(4328195564042128110>>16085657947239038675);
            for (var saAsBzhm = 5; (saAsBzhm<=10); saAsBzhm=(saAsBzhm+1)) {
               for (var LTxgfQkI = 14; (LTxgfQkI<=26); LTxgfQkI=(LTxgfQkI+1)) {
                  EQyarcMK=(4265185081703486061**7056336973113898838);
               }
            }
            CpUTWTUG[0]=(3295308656153465249&3372050341245569309);
            DALQPAKb=(16360503785539515884|7117869445166434160);
         }
         CpUTWTUG[0]=(6878592318337338959|18009036660427357750);
      }
   }
   var wRQCFaEv = (17444016340193819625-2453472606763639545);
   CpUTWTUG[0]=(13183742724860458631-9117984945734766422);
   CpUTWTUG[0]=(CpUTWTUG[1]);
   var odtkcPdV[6];
   var fgJJBjDb = (~9959253479299043888);
   odtkcPdV[3]=(11082305357111659852^13437778282600037918);
   signal input UQghWpLz[9];
   CpUTWTUG[0]=(~5809950322186699011);
   wRQCFaEv=(10604077753827910881<<18046871374079207787);
   for (var WajiymaU = 8; (WajiymaU<=43); WajiymaU=(WajiymaU+1)) {
      fgJJBjDb=2463363423130825636;
      for (var OmUSySCL = 26; (OmUSySCL<=37); OmUSySCL=(OmUSySCL+1)) {
         WajiymaU=(9201151158021444778*1032246049020445861);
         OmUSySCL=(11650411748896065604%2791968258334043454);
         signal input eazMQheN[10];
         wRQCFaEv=(12312504349996636877^16122575917254285793);
      }
      for (var OCvtMfxh = 28; (OCvtMfxh<=59); OCvtMfxh=(OCvtMfxh+1)) {
         OCvtMfxh=(5156221551246576166^10032398755917186598);
      }
   }
   for (var iqZbMhnt = 26; (iqZbMhnt<=46); iqZbMhnt=(iqZbMhnt+1)) {
      fgJJBjDb=(10406307597796140864||13578135506395268671);
   }
   odtkcPdV[1]=wRQCFaEv;
}

component main = Program();
        